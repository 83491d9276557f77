"""Profiling aid: A/B builds of the kernel library under variants/ (git-ignored; they travel to the GPU box with
the snapshot).  Nothing built here is shipped: the package only ever loads jssenv_amd/libjss_hip.so unless
JSSENV_AMD_LIB points somewhere else.

    variants/profiling.so   -DJSS_PROFILING: exports jss_profiling_set (phase ablation, LDS padding)
    variants/occ7.so        -DJSS_WAVE_MIN_BLOCKS=7: wave-per-env step kernels at 7 waves/SIMD (SGPR budget 102)
    variants/pg8.so         -DJSS_PACKED_GLOBAL_MIN_BLOCKS=8: packed per-env-table step kernels at 8 waves/SIMD (2 VGPRs in scratch)
    variants/w2occ8.so      -DJSS_WAVE2_MIN_BLOCKS=8: the two-jobs-per-lane one-step rollout at 8 waves/SIMD
    variants/nodeepwalk.so  -DJSS_EXP_NO_DEEP_WALK: the look-ahead walk without op-table reads (WRONG results; a ceiling)
    variants/plaincounters.so  -DJSS_COUNTERS_PLAIN: packed fused rollouts load / store their env's counter row with the state
                               instead of bumping it with atomics (round 6 A/B); nocounters.so: no counters at all (a ceiling)

    variants/mstep6.so, mstep7.so   -DJSS_MULTI_STEP_MIN_BLOCKS=6 / 7: the fused grid's step kernel declared for 6 / 7 wavefronts per SIMD
    variants/ptraj5.so, ptraj6.so   the packed recorders (kTraj / kSteps) declared for 5 / 6 (6: 10-23 VGPRs in scratch)
    (any other -D of the sources, e.g. -DJSS_EXP_MULTI_ONLY=<flavour>: the fused grid with one body, for per-body register
     counts -- jssenv_amd.build.build_extension(force=True, extra=[...], out=...))

One-off experiments whose findings are recorded in profiles/README.md (streaming hints on other streams, rewriting
every record, the walk without op table reads) were compile-time variants of the same sources at the commits named
there; they are not kept in the tree.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jssenv_amd.build import build_extension  # noqa: E402

VARIANTS = {"profiling": ["-DJSS_PROFILING"], "occ7": ["-DJSS_WAVE_MIN_BLOCKS=7"], "pg8": ["-DJSS_PACKED_GLOBAL_MIN_BLOCKS=8"],
            "nodeepwalk": ["-DJSS_EXP_NO_DEEP_WALK"], "w2occ8": ["-DJSS_WAVE2_MIN_BLOCKS=8"],
            "plaincounters": ["-DJSS_COUNTERS_PLAIN"], "nocounters": ["-DJSS_EXP_NO_COUNTERS"],
            "mstep6": ["-DJSS_MULTI_STEP_MIN_BLOCKS=6"], "mstep7": ["-DJSS_MULTI_STEP_MIN_BLOCKS=7"],
            "ptraj5": ["-DJSS_PTRAJ_LDS_MIN_BLOCKS=5", "-DJSS_PTRAJ_GLOBAL_MIN_BLOCKS=5"],
            "ptraj6": ["-DJSS_PTRAJ_LDS_MIN_BLOCKS=6", "-DJSS_PTRAJ_GLOBAL_MIN_BLOCKS=6"]}

if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "variants"), exist_ok=True)
    for tag in (sys.argv[1:] or VARIANTS):
        out = build_extension(force=True, extra=VARIANTS[tag], out=os.path.join(ROOT, "variants", f"{tag}.so"))
        print("built", out)
