"""Where a wavefront of the one-wavefront-per-env kernels spends its life: shader-clock stamps at the phase boundaries of
jss_kernel<*, kRollout1, *> (instrumented build: `python tools/build_instrumented.py profiling`, JSS_STAMP in the sources),
per wave, for BASELINE config 4's share (8 192 envs = one round of resident waves), all of config 4 on one GPU and config 5
padded.  Prints the median / mean cycles of every phase and of the whole life, per launch form -- one env per wavefront, and
(round 6) two envs per wavefront, the first and the second env of a wavefront apart: slot 0 of both is the wavefront's entry,
so the second env's "header words" phase is everything up to the moment it claims its records, the first env's step included.

    JSSENV_AMD_LIB=$PWD/variants/profiling.so python tools/gpu_wave_timeline.py          (GPU box)
"""
import ctypes as C
import os
import sys

os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from jssenv_amd import BatchedJssEnv, builtin_instance  # noqa: E402
from jssenv_amd.instances import synthetic_packed  # noqa: E402

dev = torch.device("cuda", 0)
PHASES = [("header words (scalar loads)", 0, 1), ("state loads + unpack", 1, 2), ("policy (select_action)", 2, 3),
          ("step: allocate + event jump", 3, 7), ("step: _prioritization_non_final", 7, 8), ("step: _check_no_op", 8, 9),
          ("step: settle (op-table refill)", 9, 4), ("pack + state / mask stores issued", 4, 5), ("observation (LDS image, stores)", 5, 6),
          ("WHOLE LIFE", 0, 6)]

FORMS = sys.argv[1:] or ["auto-1env", "auto-2env"]
for label, src, batch, kw in [(f"{l} [{k}]", s_, b_, dict(kw_, kernel=k)) for l, s_, b_, kw_ in (
        ("config 4 share: synthetic 50x20 x 8192", lambda: synthetic_packed(8192, 50, 20), 8192, {}),
        ("config 4 whole: synthetic 50x20 x 65536", lambda: synthetic_packed(65536, 50, 20), 65536, {}),
        ("config 5 padded: mixed ta01-80 x 32768, env i <- ta(1 + i % 80)", lambda: [builtin_instance(f"ta{k:02d}") for k in range(1, 81)], 32768,
         dict(order="interleaved"))) for k in FORMS if not (k.endswith("2env") and l.startswith("config 5"))]:
    env = BatchedJssEnv(src(), batch=batch, device=dev, seed=0, **kw)
    lib = env.backend.lib
    if not hasattr(lib, "jss_profiling_stamps"):
        raise SystemExit("needs the instrumented build: JSSENV_AMD_LIB=variants/profiling.so")
    lib.jss_profiling_stamps.argtypes, lib.jss_profiling_stamps.restype = [C.c_void_p], C.c_int
    env.reset()
    env.rollout("random", n_iter=170)                          # mid-episode state
    stamps = torch.zeros((batch, 16), dtype=torch.int64, device=dev)
    acc, raw_all = [], []
    for rep in range(6):
        lib.jss_profiling_stamps(stamps.data_ptr())
        env.rollout("random", n_iter=1)
        torch.cuda.synchronize()
        lib.jss_profiling_stamps(None)
        s = stamps.cpu().numpy().astype(np.int64)
        ok = (s[:, 6] > s[:, 0]) & (s[:, 4] > 0)                 # waves that stepped (an env found done is reset instead: no stamps 3..9)
        acc.append(s[ok])
        raw_all.append(s)
        stamps.zero_()
        env.rollout("random", n_iter=3)
    two = kw["kernel"].endswith("2env")
    groups = [("", np.concatenate(acc))]
    if two:                                                      # envs 2w / 2w + 1 of wavefront w: first / second in turn
        idx = [np.flatnonzero((s[:, 6] > s[:, 0]) & (s[:, 4] > 0)) for s in raw_all]
        groups = [(" -- the wavefront's FIRST env", np.concatenate([s[i[i % 2 == 0]] for s, i in zip(raw_all, idx)])),
                  (" -- the wavefront's SECOND env (slot 0 = the wavefront's entry)", np.concatenate([s[i[i % 2 == 1]] for s, i in zip(raw_all, idx)]))]
    for suffix, s in groups:
        print(f"== {label}{suffix}: {len(s)} env lives ==")
        for name, a, b in PHASES:
            d = (s[:, b] - s[:, a]).astype(np.float64)
            print(f"  {name:42s} median {np.median(d):8.0f}  mean {d.mean():8.0f}  p90 {np.percentile(d, 90):8.0f} cycles")
    if two:                                                      # the wavefront: entry of the pair -> end of its second env
        life = np.concatenate([(s[1::2, 6] - s[0::2, 0])[(s[1::2, 6] > s[0::2, 0]) & (s[1::2, 4] > 0) & (s[0::2, 4] > 0)] for s in raw_all]).astype(np.float64)
        print(f"  {'WHOLE WAVEFRONT (two envs)':42s} median {np.median(life):8.0f}  mean {life.mean():8.0f}  p90 {np.percentile(life, 90):8.0f} cycles")
    del env, stamps
