"""profiles/hbm_traffic.json from the rocprofv3 PMC summaries under profiles/<round>_*/ (tools/gpu_profile.sh ->
tools/summarize_prof.py).  bench.py prints these static numbers as roofline.traffic; every entry carries the sha256 of
the kernel sources it was measured on (bench.py: csrc_hash), so a stale entry is detectable.

    python tools/make_traffic_json.py r03
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import b_alg, csrc_hash  # noqa: E402

# key -> (profile dir suffix, kernel substring, kernel label, algorithmic bytes per env step, batch)
MIXED_ALG = None
WORKLOADS = {
    "ta01_b65536": ("ta01_single", "jss_packed_kernel<16, 5, 2>", "jss_packed_kernel<16,kRollout1,kTabLdsC>", b_alg(15, 15), 65536),
    "ta01_b262144": ("ta01_b262144", "jss_packed_kernel<16, 5, 2>", "jss_packed_kernel<16,kRollout1,kTabLdsC>", b_alg(15, 15), 262144),
    "ta01_b4096": ("ta01_b4096", "jss_packed_kernel<16, 5, 2>", "jss_packed_kernel<16,kRollout1,kTabLdsC>", b_alg(15, 15), 4096),
    "syn15x15_b65536": ("syn15x15", "jss_packed_kernel<16, 5, 3>", "jss_packed_kernel<16,kRollout1,kTabGlobalM> (24-byte medium records)", b_alg(15, 15), 65536),
    "ta41_b16384": ("ta41", "jss_packed_kernel<32, 5, 2>", "jss_packed_kernel<32,kRollout1,kTabLdsC>", b_alg(30, 20), 16384),
    "syn50x20_b8192": ("syn50x20", "jss_kernel<1, 5, 3>", "jss_kernel<1,kRollout1,kTabGlobalM> (24-byte medium records)", b_alg(50, 20), 8192),
    "mixed_b32768": ("mixed", "jss_kernel<2, 5, 1>", "jss_kernel<2,kRollout1,kTabGlobal> (order='interleaved'; one-job-per-lane body for J <= 64)", None, 32768),
    "mixed_bucketed_b32768": ("mixed_bucketed", "jss_multi_kernel<5>", "jss_multi_kernel<kRollout1>: one grid over the four shape classes", None, 32768),
    "mixed_by_shape_b32768": ("mixed_by_shape", "jss_multi_kernel<5>", "jss_multi_kernel<kRollout1>: class-specialised bodies on the padded rows", None, 32768),
    "syn50x20_b65536": ("syn50x20_b65536", "jss_kernel_two<5, 3>", "jss_kernel_two<kRollout1,kTabGlobalM> (two envs per wavefront, one after the other; medium records)", b_alg(50, 20), 65536),
}


def counter(path, kernel, name):
    for r in csv.DictReader(open(path)):
        if r["kernel"] == kernel and r["counter"] == name:
            return float(r["median_per_dispatch"])
    raise KeyError((path, kernel, name))


def main(rnd):
    from jssenv_amd import builtin_instance
    mixed = sum(b_alg(i.jobs, i.machines) for i in (builtin_instance(f"ta{k:02d}") for k in range(1, 81))) / 80.0
    out = {"_note": "HBM-side bytes per launch of the benchmarked kernels from rocprofv3 PMC passes (separate --pmc FETCH_SIZE and "
                    "--pmc WRITE_SIZE runs of `bench.py --launch eager`, tools/gpu_profile.sh; summaries under profiles/" + rnd +
                    "_*).  FETCH_SIZE and WRITE_SIZE are in KiB; bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE is doubled "
                    "per the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md (HBM section; exact for the wide coalesced "
                    "state loads, an over-count for the narrow op-table gathers of the per-env-table kernels).  The counters sit on "
                    "the L2's fabric side and include Infinity-Cache hits.  Static: bench.py prints these numbers, it does not "
                    "re-measure them; csrc_sha16 = sha256 of jssenv_amd/csrc/* + include/jss_hip.h at measurement time."}
    sha = csrc_hash()
    for key, (suffix, kern, label, alg, batch) in WORKLOADS.items():
        d = os.path.join(ROOT, "profiles", f"{rnd}_{suffix}")
        try:
            fetch = counter(os.path.join(d, "pmc_fetch_summary.csv"), kern, "FETCH_SIZE")
            write = counter(os.path.join(d, "pmc_write_summary.csv"), kern, "WRITE_SIZE")
        except (OSError, KeyError) as exc:
            print("skip", key, exc)
            continue
        a = (alg if alg is not None else mixed) * batch
        b = (2 * fetch + write) * 1024
        out[key] = {"kernel": label, "fetch_size_kib": fetch, "write_size_kib": write, "bytes_per_launch": int(b),
                    "algorithmic_bytes_per_launch_if_every_env_steps": int(a), "ratio": round(b / a, 3),
                    "source": f"{rnd}_{suffix}/pmc_fetch_summary.csv + pmc_write_summary.csv", "round": rnd, "csrc_sha16": sha}
        try:      # wave-cycles per env step and the share of them spent waiting (SQ_WAVE_CYCLES, SQ_WAIT_ANY)
            pm = os.path.join(d, "pmc_pmc1_summary.csv")
            cyc, wait, waves = counter(pm, kern, "SQ_WAVE_CYCLES"), counter(pm, kern, "SQ_WAIT_ANY"), counter(pm, kern, "SQ_WAVES")
            out[key].update(sq_wave_cycles_per_launch=cyc, sq_wait_any_per_launch=wait, sq_waves_per_launch=waves,
                            wave_cycles_per_env_step=round(cyc / batch, 1), wait_fraction=round(wait / cyc, 3),
                            valu_per_wave=round(counter(pm, kern, "SQ_INSTS_VALU") / waves, 1),
                            salu_per_wave=round(counter(pm, kern, "SQ_INSTS_SALU") / waves, 1))
        except (OSError, KeyError, ZeroDivisionError) as exc:
            print("no pmc1 for", key, exc)
    with open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps({k: v.get("ratio") for k, v in out.items() if k != "_note"}))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r03")
