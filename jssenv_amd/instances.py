"""Job-shop instances: text parser, Taillard generator, packed op tables.

Replaces the parsing half of ``JssEnv.__init__`` (reference
JSSEnv/envs/jss_env.py:72-95): line 1 is ``J M``; each of the next J lines
holds M ``machine duration`` pairs (machines 0-indexed).  The reference keeps
an ``(J, M, 2)`` int64 ``instance_matrix``; the device path wants one int32
per operation, ``machine << 16 | duration``, padded to ``(Jmax, Mmax)`` so a
batch of ragged instances is one contiguous tensor.

The per-instance constants the observation is normalised with are computed
here exactly as the reference does (jss_env.py:86-89):
``max_time_op`` = longest single operation, ``max_time_jobs`` = longest job
(sum over its ops), ``sum_op`` = sum of all durations.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Sequence, Union

import numpy as np

MAX_JOBS = 128       # two jobs per lane of a 64-wide wavefront
MAX_MACHINES = 64    # machine m lives on lane m
MAX_DURATION = 0xFFFF
OP_MACHINE_SHIFT = 16
INST_RECORD_INTS = 12   # JSS_NI

_DATA_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "instances.npz")


@dataclass(frozen=True)
class Instance:
    """One job-shop instance. ``machine[j, k]`` / ``duration[j, k]`` describe op k of job j."""

    name: str
    machine: np.ndarray   # (J, M) int32
    duration: np.ndarray  # (J, M) int32

    def __post_init__(self):
        m, d = self.machine, self.duration
        if m.ndim != 2 or m.shape != d.shape:
            raise ValueError("machine/duration must be (J, M) arrays of the same shape")
        J, M = m.shape
        # same sanity checks as jss_env.py:91-95
        if J < 1:
            raise ValueError("instance needs at least one job")
        if M < 2:
            raise ValueError("We need at least 2 machines")
        if J > MAX_JOBS or M > MAX_MACHINES:
            raise ValueError(f"instance {J}x{M} exceeds the device limits {MAX_JOBS}x{MAX_MACHINES}")
        if m.min() < 0 or m.max() >= M:
            raise ValueError("machine index out of range")
        # Zero-length ops would put an event at the current time with no busy
        # machine; the event-queue == busy-machines equivalence the kernels
        # rely on needs d >= 1 (true for every shipped and Taillard instance).
        if d.min() < 1 or d.max() > MAX_DURATION:
            raise ValueError("durations must be in [1, 65535]")

    @property
    def jobs(self) -> int:
        return int(self.machine.shape[0])

    @property
    def machines(self) -> int:
        return int(self.machine.shape[1])

    @property
    def max_time_op(self) -> int:          # jss_env.py:86
        return int(self.duration.max())

    @property
    def jobs_length(self) -> np.ndarray:   # jss_env.py:87
        return self.duration.sum(axis=1)

    @property
    def max_time_jobs(self) -> int:        # jss_env.py:89
        return int(self.jobs_length.max())

    @property
    def sum_op(self) -> int:               # jss_env.py:88
        return int(self.duration.sum())

    @property
    def instance_matrix(self) -> np.ndarray:
        """(J, M, 2) int64 array with the reference's layout (jss_env.py:78,85)."""
        return np.stack([self.machine, self.duration], axis=-1).astype(np.int64)

    def packed(self, jmax: int | None = None, mmax: int | None = None) -> np.ndarray:
        """(jmax, mmax) int32 table of ``machine << 16 | duration`` (0 = padding)."""
        J, M = self.machine.shape
        jmax = J if jmax is None else jmax
        mmax = M if mmax is None else mmax
        out = np.zeros((jmax, mmax), dtype=np.int32)
        out[:J, :M] = (self.machine.astype(np.int32) << OP_MACHINE_SHIFT) | self.duration.astype(np.int32)
        return out

    def to_text(self) -> str:
        J, M = self.machine.shape
        lines = [f"{J} {M}"]
        for j in range(J):
            lines.append(" ".join(f"{int(self.machine[j, k])} {int(self.duration[j, k])}" for k in range(M)))
        return "\n".join(lines) + "\n"


def parse_instance_text(text: str, name: str = "") -> Instance:
    rows = [ln.split() for ln in text.splitlines()]
    if not rows or len(rows[0]) != 2:
        raise ValueError("first line must be 'J M'")
    J, M = int(rows[0][0]), int(rows[0][1])
    body = rows[1:]
    if len(body) < J:
        raise ValueError(f"expected {J} job lines, found {len(body)}")
    machine = np.zeros((J, M), dtype=np.int32)
    duration = np.zeros((J, M), dtype=np.int32)
    for j in range(J):
        vals = [int(v) for v in body[j]]
        # the reference asserts exactly M (machine, time) pairs per line (jss_env.py:81)
        if len(vals) != 2 * M:
            raise ValueError(f"job line {j} holds {len(vals)} integers, expected {2 * M}")
        machine[j] = vals[0::2]
        duration[j] = vals[1::2]
    for extra in body[J:]:
        if extra:
            raise ValueError("trailing non-empty lines after the last job")
    return Instance(name, machine, duration)


def load_instance_file(path: Union[str, os.PathLike]) -> Instance:
    with open(path, "r") as fh:
        return parse_instance_text(fh.read(), name=os.path.basename(str(path)))


# --------------------------------------------------------------------------
# Taillard (1993) generator: Lehmer LCG a=16807 mod 2^31-1 via Schrage.
# --------------------------------------------------------------------------
_LCG_A, _LCG_B, _LCG_C, _LCG_M = 16807, 127773, 2836, 2147483647


class _TaillardRng:
    def __init__(self, seed: int):
        if not 0 < seed < _LCG_M:
            raise ValueError("seed must be in [1, 2^31-2]")
        self.x = int(seed)

    def unif(self, low: int, high: int) -> int:
        k = self.x // _LCG_B
        self.x = _LCG_A * (self.x % _LCG_B) - k * _LCG_C
        if self.x < 0:
            self.x += _LCG_M
        return low + int((self.x / _LCG_M) * (high - low + 1))


def taillard_instance(jobs: int, machines: int, time_seed: int, machine_seed: int, name: str = "") -> Instance:
    """Taillard job-shop instance: durations U{1..99}, machine order by LCG shuffle.

    ``taillard_instance(15, 15, 840612802, 398197754)`` is ta01.
    """
    rt = _TaillardRng(time_seed)
    duration = np.zeros((jobs, machines), dtype=np.int32)
    for j in range(jobs):
        for k in range(machines):
            duration[j, k] = rt.unif(1, 99)
    rm = _TaillardRng(machine_seed)
    machine = np.tile(np.arange(machines, dtype=np.int32), (jobs, 1))
    for j in range(jobs):
        for k in range(machines):
            s = rm.unif(k, machines - 1)
            machine[j, k], machine[j, s] = machine[j, s], machine[j, k]
    return Instance(name or f"tai_{jobs}x{machines}_{time_seed}_{machine_seed}", machine, duration)


def synthetic_batch(n: int, jobs: int, machines: int, first: int = 0) -> List[Instance]:
    """BASELINE config 4 rule: env i uses time_seed = 1 + 2i, machine_seed = 2 + 2i."""
    return [taillard_instance(jobs, machines, 1 + 2 * (first + i), 2 + 2 * (first + i)) for i in range(n)]


def _lcg_unif(x: np.ndarray, low, high) -> np.ndarray:
    """One Taillard LCG draw per element of the state vector ``x`` (advanced in place)."""
    k = x // _LCG_B
    x[:] = _LCG_A * (x % _LCG_B) - k * _LCG_C
    x[x < 0] += _LCG_M
    return low + ((x / _LCG_M) * (high - low + 1)).astype(np.int64)


def synthetic_arrays(n: int, jobs: int, machines: int, first: int = 0):
    """The instances of ``synthetic_batch(n, jobs, machines, first)`` as two (n, J, M) int32 arrays
    (machine, duration), generated for all n instances at once (every instance has its own LCG streams, so the
    J*M draws vectorise over the instance axis).  65 536 instances of 15x15 take about a second."""
    idx = first + np.arange(n, dtype=np.int64)
    xt, xm = 1 + 2 * idx, 2 + 2 * idx
    if n and (xm.max() >= _LCG_M):
        raise ValueError("seed must be in [1, 2^31-2]")
    duration = np.zeros((n, jobs, machines), dtype=np.int32)
    for j in range(jobs):
        for k in range(machines):
            duration[:, j, k] = _lcg_unif(xt, 1, 99)
    machine = np.tile(np.arange(machines, dtype=np.int32), (n, jobs, 1))
    rows = np.arange(n)
    for j in range(jobs):
        for k in range(machines):
            s = _lcg_unif(xm, k, machines - 1)
            a, b = machine[rows, j, k].copy(), machine[rows, j, s].copy()
            machine[rows, j, k], machine[rows, j, s] = b, a
    return machine, duration


def synthetic_packed(n: int, jobs: int, machines: int, first: int = 0) -> "PackedBatch":
    """``pack_batch(synthetic_batch(...))`` without building n Instance objects (bench-sized batches)."""
    machine, duration = synthetic_arrays(n, jobs, machines, first)
    ops = ((machine << OP_MACHINE_SHIFT) | duration).astype(np.int32)
    rem = np.cumsum(duration[:, :, ::-1], axis=2)[:, :, ::-1].astype(np.int32)
    jl = duration.sum(axis=2)
    mto, mtj, sop = duration.max(axis=(1, 2)), jl.max(axis=1), jl.sum(axis=1)
    rec = np.zeros((n, INST_RECORD_INTS), dtype=np.int32)
    rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3], rec[:, 4] = jobs, machines, mto, mtj, sop
    vals = np.stack([mto, mtj, sop, np.full(n, machines)], axis=1).astype(np.float32)
    rec[:, 5:9] = (np.float32(1.0) / vals).view(np.int32)
    as32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)  # noqa: E731
    return PackedBatch(ops=np.ascontiguousarray(ops), rem=np.ascontiguousarray(rem), inst=rec,
                       jobs=np.full(n, jobs, dtype=np.int32), machines=np.full(n, machines, dtype=np.int32),
                       max_time_op=as32(mto), max_time_jobs=as32(mtj), sum_op=as32(sop), jmax=jobs, mmax=machines)


# --------------------------------------------------------------------------
# Shipped benchmark set (ta01-ta80, dmu16-dmu20), stored as packed arrays.
# --------------------------------------------------------------------------
_cache = {}


def _bundle():
    if "npz" not in _cache:
        if not os.path.isfile(_DATA_FILE):
            raise FileNotFoundError(f"{_DATA_FILE} missing; run tools/pack_instances.py")
        with np.load(_DATA_FILE) as z:
            _cache["npz"] = {k: z[k] for k in z.files}
    return _cache["npz"]


def available_instances() -> List[str]:
    return [str(n) for n in _bundle()["names"]]


def builtin_instance(name: str) -> Instance:
    b = _bundle()
    names = [str(n) for n in b["names"]]
    if name not in names:
        raise KeyError(f"unknown instance {name!r}")
    i = names.index(name)
    J, M = int(b["shape"][i, 0]), int(b["shape"][i, 1])
    off = int(b["offset"][i])
    ops = b["ops"][off: off + J * M].reshape(J, M).astype(np.int32)
    return Instance(name, ops >> OP_MACHINE_SHIFT, ops & MAX_DURATION)


def resolve_instance(spec) -> Instance:
    """Accept an Instance, a builtin name ('ta01') or a path to a Taillard-format text file."""
    if isinstance(spec, Instance):
        return spec
    s = str(spec)
    if os.path.isfile(s):
        return load_instance_file(s)
    base = os.path.basename(s)
    try:
        return builtin_instance(base)
    except (KeyError, FileNotFoundError):
        raise FileNotFoundError(f"instance {spec!r} is neither a file nor a builtin instance name")


@dataclass
class PackedBatch:
    """Host-side description of a batch of instances, ready to upload (include/jss_hip.h JssDesc)."""

    ops: np.ndarray            # (n_tables, Jmax, Mmax) int32, machine<<16|duration
    rem: np.ndarray            # (n_tables, Jmax, Mmax) int32, rem[j][k] = sum of the durations of ops k..M-1 of job j
    inst: np.ndarray           # (n_tables, NI) int32 instance records (JSS_I_*: J, M, normalisers, float32 reciprocals)
    jobs: np.ndarray           # (n_tables,) int32
    machines: np.ndarray       # (n_tables,) int32
    max_time_op: np.ndarray    # (n_tables,) int32
    max_time_jobs: np.ndarray  # (n_tables,) int32
    sum_op: np.ndarray         # (n_tables,) int32
    jmax: int
    mmax: int


def instance_record(inst: Instance) -> np.ndarray:
    """The JSS_I_* record of one instance: J, M, the observation's normalisers (jss_env.py:86-89) and their
    correctly rounded float32 reciprocals (the kernels divide as q = a * r, one residual correction)."""
    rec = np.zeros(INST_RECORD_INTS, dtype=np.int32)
    vals = (inst.max_time_op, inst.max_time_jobs, inst.sum_op, inst.machines)
    rec[0:5] = (inst.jobs, inst.machines, inst.max_time_op, inst.max_time_jobs, inst.sum_op)
    rec[5:9] = (np.float32(1.0) / np.asarray(vals, dtype=np.float32)).view(np.int32)
    return rec


def pack_batch(instances: Sequence[Instance], jmax: int | None = None, mmax: int | None = None) -> PackedBatch:
    jmax = max(i.jobs for i in instances) if jmax is None else jmax
    mmax = max(i.machines for i in instances) if mmax is None else mmax
    n = len(instances)
    ops = np.zeros((n, jmax, mmax), dtype=np.int32)
    rem = np.zeros((n, jmax, mmax), dtype=np.int32)
    rec = np.zeros((n, INST_RECORD_INTS), dtype=np.int32)
    for i, inst in enumerate(instances):
        ops[i] = inst.packed(jmax, mmax)
        rem[i, :inst.jobs, :inst.machines] = np.cumsum(inst.duration[:, ::-1], axis=1)[:, ::-1]
        rec[i] = instance_record(inst)
    as32 = lambda it: np.asarray(list(it), dtype=np.int32)  # noqa: E731
    return PackedBatch(
        ops=ops, rem=rem, inst=rec,
        jobs=as32(i.jobs for i in instances),
        machines=as32(i.machines for i in instances),
        max_time_op=as32(i.max_time_op for i in instances),
        max_time_jobs=as32(i.max_time_jobs for i in instances),
        sum_op=as32(i.sum_op for i in instances),
        jmax=jmax,
        mmax=mmax,
    )


# --------------------------------------------------------------------------
# Packed on-disk batch format: one .npz holding the upload-ready tables.
# --------------------------------------------------------------------------
_BATCH_FORMAT = 1


def save_batch(path, batch: "PackedBatch") -> None:
    """Write a PackedBatch as one .npz (arrays only, no pickling): the op tables as uploaded, the instance
    records, and enough per-table scalars to rebuild everything else."""
    with open(path, "wb") as fh:
        np.savez(fh, format=np.int32(_BATCH_FORMAT), ops=batch.ops, inst=batch.inst)


def load_batch(path) -> "PackedBatch":
    """Read a file written by save_batch; derived tables (remaining work, reciprocals) are recomputed and the
    stored instance records are checked against them."""
    with np.load(path, allow_pickle=False) as z:
        if int(z["format"]) != _BATCH_FORMAT:
            raise ValueError(f"unknown packed batch format {int(z['format'])}")
        ops, inst = np.ascontiguousarray(z["ops"], dtype=np.int32), np.ascontiguousarray(z["inst"], dtype=np.int32)
    if ops.ndim != 3 or inst.shape != (ops.shape[0], INST_RECORD_INTS):
        raise ValueError("malformed packed batch")
    n, jmax, mmax = ops.shape
    insts = []
    for i in range(n):
        J, M = int(inst[i, 0]), int(inst[i, 1])
        if not (1 <= J <= jmax and 2 <= M <= mmax):
            raise ValueError(f"table {i}: bad shape {J}x{M}")
        insts.append(Instance(f"table{i}", ops[i, :J, :M] >> OP_MACHINE_SHIFT, ops[i, :J, :M] & MAX_DURATION))
    pk = pack_batch(insts, jmax, mmax)
    if not (np.array_equal(pk.ops, ops) and np.array_equal(pk.inst, inst)):
        raise ValueError("packed batch is inconsistent (padding or instance records do not match the op tables)")
    return pk
