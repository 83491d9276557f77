"""ctypes mirror of include/jss_hip.h (structs, constants, prototypes).

Pure declarations: no torch, no device access.  ``bind(lib)`` attaches the
prototypes to a loaded ``libjss_hip.so`` (or its host-core twin ``libjss_cpu.so``:
identical symbols) and fails loudly if a symbol the header declares is missing.
"""
from __future__ import annotations

import ctypes as C
import os

ABI_VERSION = 10
STATE_LAYOUT = 7     # version of the state tensors' layout (checkpoints): unchanged since ABI v7
MAX_JOBS, MAX_MACHINES = 128, 64
F_TODO, F_CUR, F_LEFT, F_PERF, F_IDLE, F_IDLE_LAST, F_F4, F_NEXT, NF = 0, 1, 2, 3, 4, 5, 6, 7, 8
TODO_MASK, FLAG_LEGAL, FLAG_BLOCKED, NEXT2_SHIFT = 255, 256, 512, 10
# compact 16-byte record of shared-instance batches (JSS_FC_*): no cached ops, packed words
FC_W0, FC_LEFT_F4, FC_IDLE, FC_IDLE_LAST, NFC = 0, 1, 2, 3, 4
FC_TODO_MASK, FC_FLAG_LEGAL, FC_FLAG_BLOCKED, FC_FLAG_F4_ONE, FC_PERF_SHIFT = 127, 128, 256, 512, 10
# medium 24-byte record of per-env-instance batches with jobs, machines <= 32 (JSS_FM_*): three 21-bit cached ops, no machine clocks
FM_W0, FM_LEFT_F4, FM_PERF_NEXT, FM_NEXT_NEXT2, FM_IDLE, FM_IDLE_LAST, NFM = 0, 1, 2, 3, 4, 5, 6
FM_TODO_MASK, FM_FLAG_LEGAL, FM_FLAG_BLOCKED, FM_FLAG_F4_ONE, FM_CUR_SHIFT, FM_OP_MASK = 63, 64, 128, 256, 9, 0x1FFFFF
H_CLOCK, H_EPISODE, H_STEP, H_STATUS = 0, 1, 2, 3
NH = 4
C_JOBS, C_MACHINES, C_MAX_TIME_OP, C_TABLE, C_MAX_TIME_JOBS, C_SUM_OP = 0, 1, 2, 3, 4, 5
C_RCP_MAX_TIME_OP, C_RCP_MAX_TIME_JOBS, C_RCP_SUM_OP, C_RCP_MACHINES, NC = 6, 7, 8, 9, 12
STATUS_NOOP = 256
F4_ONE = -1
I_JOBS, I_MACHINES, I_MAX_TIME_OP, I_MAX_TIME_JOBS, I_SUM_OP = 0, 1, 2, 3, 4
I_RCP_MAX_TIME_OP, I_RCP_MAX_TIME_JOBS, I_RCP_SUM_OP, I_RCP_MACHINES, NI = 5, 6, 7, 8, 12
ERR_ILLEGAL_ACTION, ERR_NOPE_IDLE, ERR_BAD_ACTION = 1, 2, 4
ACTION_SKIP, ACTION_RESET, ACTION_CLOSE = -1, -2, -3
POLICY = {"random": 0, "FIFO": 1, "SPT": 2, "MWR": 3, "LWR": 4, "MOR": 5, "LOR": 6, "CR": 7}
ROLLOUT_AUTORESET, ROLLOUT_FORK_JOIN = 1, 2


def cr_kind(due_date_factor: float = 1.5):
    """The `kind` code of CriticalRatio with a due-date factor other than the default (JSS_POLICY_CR_FACTOR): factor = p / q
    with q a power of two <= 64 and p <= 255, or None when the factor has no such form (the host loop serves it then)."""
    from fractions import Fraction
    f = Fraction(due_date_factor).limit_denominator(64)
    if float(f) != float(due_date_factor) or f <= 0 or f.numerator > 255 or f.denominator & (f.denominator - 1):
        return None
    if (f.numerator, f.denominator) == (3, 2):
        return POLICY["CR"]
    return POLICY["CR"] | (f.numerator << 8) | (f.denominator << 16)


POLICY_CR_F64 = POLICY["CR"] | (1 << 24)    # JSS_POLICY_CR_F64: the factor travels as the double JssDesc.cr_factor


def policy_code(kind):
    """str | int -> the int the C ABI takes."""
    return POLICY[kind] if isinstance(kind, str) else int(kind)
# JssDesc.kernel: "wave" forces one wavefront per env; the "...-1env" forms add JSS_KERNEL_ONE_ENV_PER_WAVE (a wavefront of the
# one-step launches never serves two envs in turn: A/B runs, tests)
# ("...-2env": JSS_KERNEL_TWO_ENVS_PER_WAVE, they always do -- tests on small batches)
KERNEL = {"auto": 0, "wave": 1, "auto-1env": 2, "wave-1env": 3, "auto-2env": 4, "wave-2env": 5}
E_NULL, E_SHAPE, E_KIND, E_LDS, E_RESIDENT, E_SESSION = -1, -2, -3, -4, -5, -6
MAX_SUB_BATCHES = 16

SYMBOLS = ("jss_abi_version", "jss_error_string", "jss_backend", "jss_reset", "jss_step", "jss_advance", "jss_policy",
           "jss_rollout", "jss_rollout_steps", "jss_rollout_steps_multi", "jss_trajectory", "jss_sync_check",
           "jss_step_autoreset", "jss_policy_step_steps", "jss_steps", "jss_session_open", "jss_session_post", "jss_session_wait", "jss_session_step", "jss_session_close",
           "jss_multi_reset", "jss_multi_step", "jss_multi_policy", "jss_multi_rollout")

_p = C.c_void_p


class JssDesc(C.Structure):
    _fields_ = [("batch", C.c_int32), ("jmax", C.c_int32), ("mmax", C.c_int32), ("n_tables", C.c_int32),
                ("ops", _p), ("rem", _p), ("inst", _p), ("table_of_env", _p), ("env_ids", _p),
                ("env_id_base", C.c_int64), ("kernel", C.c_int32), ("threads", C.c_int32),
                ("jmin", C.c_int32), ("record_ints", C.c_int32), ("cr_factor", C.c_double),
                ("jclass", C.c_int32), ("mclass", C.c_int32)]


class JssState(C.Structure):
    _fields_ = [("env", _p), ("env_const", _p), ("job", _p), ("machine", _p), ("solution", _p), ("counters", _p)]


class JssOut(C.Structure):
    _fields_ = [("real_obs", _p), ("action_mask", _p), ("reward", _p), ("done", _p), ("makespan", _p)]


class JssTraj(C.Structure):
    _fields_ = [("real_obs", _p), ("action_mask", _p), ("action", _p), ("reward", _p), ("done", _p),
                ("stride", C.c_int64)]       # envs between two steps' slots (0 = the call's batch): a range of a larger batch


class JssSession(C.Structure):
    _fields_ = [("mail", _p), ("progress", _p), ("status", _p), ("depth", C.c_int32), ("timeout_ms", C.c_int32),
                ("slots", C.c_int32), ("reserved", C.c_int32)]


def library_path(name: str = "libjss_hip.so") -> str:
    """In-tree library; JSSENV_AMD_LIB points development builds at another build of the same ABI."""
    if name == "libjss_hip.so" and os.environ.get("JSSENV_AMD_LIB"):
        return os.environ["JSSENV_AMD_LIB"]
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), name)


def bind(lib):
    """Attach prototypes; raises AttributeError naming the first missing symbol."""
    for name in SYMBOLS:
        if not hasattr(lib, name):
            raise AttributeError(f"library does not export {name}")
    D, S, O = C.POINTER(JssDesc), C.POINTER(JssState), C.POINTER(JssOut)
    lib.jss_abi_version.restype, lib.jss_abi_version.argtypes = C.c_int, []
    lib.jss_error_string.restype, lib.jss_error_string.argtypes = C.c_char_p, [C.c_int]
    lib.jss_backend.restype, lib.jss_backend.argtypes = C.c_char_p, []
    lib.jss_reset.restype, lib.jss_reset.argtypes = C.c_int, [D, S, O, _p, _p]
    lib.jss_step.restype, lib.jss_step.argtypes = C.c_int, [D, S, _p, O, _p]
    lib.jss_step_autoreset.restype, lib.jss_step_autoreset.argtypes = C.c_int, [D, S, _p, O, _p]
    lib.jss_advance.restype, lib.jss_advance.argtypes = C.c_int, [D, S, _p, _p, O, _p]
    lib.jss_policy.restype, lib.jss_policy.argtypes = C.c_int, [D, S, C.c_int, C.c_uint64, C.c_uint32, _p, _p]
    lib.jss_rollout.restype = C.c_int
    lib.jss_rollout.argtypes = [D, S, O, C.c_int, C.c_uint64, C.c_uint32, C.c_int32, C.c_int32, _p]
    lib.jss_rollout_steps.restype = C.c_int
    lib.jss_rollout_steps.argtypes = [D, S, O, C.c_int, C.c_uint64, C.c_uint32, C.c_int32, C.c_int32, C.c_int32,
                                      C.POINTER(_p)]
    lib.jss_policy_step_steps.restype = C.c_int
    lib.jss_policy_step_steps.argtypes = [D, S, O, C.c_int, C.c_uint64, C.c_uint32, _p, C.c_int32, C.c_int32, C.c_int32,
                                          C.POINTER(_p)]
    lib.jss_rollout_steps_multi.restype = C.c_int
    lib.jss_rollout_steps_multi.argtypes = [C.c_int32, C.POINTER(D), C.POINTER(S), C.POINTER(O), C.c_int, C.c_uint64, C.c_uint32,
                                            C.c_int32, C.c_int32, C.POINTER(_p)]
    PD, PS, PO, PP = C.POINTER(D), C.POINTER(S), C.POINTER(O), C.POINTER(_p)
    lib.jss_multi_reset.restype, lib.jss_multi_reset.argtypes = C.c_int, [C.c_int32, PD, PS, PO, PP, _p]
    lib.jss_multi_step.restype, lib.jss_multi_step.argtypes = C.c_int, [C.c_int32, PD, PS, PP, PO, C.c_int32, _p]
    lib.jss_multi_policy.restype = C.c_int
    lib.jss_multi_policy.argtypes = [C.c_int32, PD, PS, C.c_int, C.c_uint64, C.c_uint32, PP, _p]
    lib.jss_multi_rollout.restype = C.c_int
    lib.jss_multi_rollout.argtypes = [C.c_int32, PD, PS, PO, C.c_int, C.c_uint64, C.c_uint32, C.c_int32, C.c_int32, C.c_int32, PP]
    lib.jss_trajectory.restype = C.c_int
    lib.jss_trajectory.argtypes = [D, S, O, C.POINTER(JssTraj), C.c_int, C.c_uint64, C.c_uint32, C.c_int32, C.c_int32, _p]
    lib.jss_sync_check.restype, lib.jss_sync_check.argtypes = C.c_int, [_p]
    SS = C.POINTER(JssSession)
    lib.jss_steps.restype, lib.jss_steps.argtypes = C.c_int, [D, S, O, C.POINTER(JssTraj), _p, C.c_int32, _p]
    lib.jss_session_open.restype, lib.jss_session_open.argtypes = C.c_int, [D, S, O, SS, _p]
    lib.jss_session_post.restype = C.c_int
    lib.jss_session_post.argtypes = [D, SS, _p, C.c_int32, C.c_int32, C.c_int32, _p]
    lib.jss_session_wait.restype, lib.jss_session_wait.argtypes = C.c_int, [D, SS, C.c_int32, _p]
    lib.jss_session_close.restype, lib.jss_session_close.argtypes = C.c_int, [D, SS, C.c_int32, _p]
    lib.jss_session_step.restype, lib.jss_session_step.argtypes = C.c_int, [D, SS, _p, C.c_int32, _p]
    if lib.jss_abi_version() != ABI_VERSION:
        raise RuntimeError(f"library ABI {lib.jss_abi_version()} != expected {ABI_VERSION}")
    return lib


def check(lib, rc: int, what: str):
    if rc != 0:
        msg = lib.jss_error_string(rc)
        raise RuntimeError(f"{what} failed: {rc} ({msg.decode() if msg else '?'})")
