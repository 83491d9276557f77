"""The per-round table of profiles/README.md, computed from the committed rocprofv3 summaries -- never typed.

    python tools/profile_table.py r05          # prints the markdown table
    python tools/profile_table.py r05 --write  # and replaces the block between the r05 markers in profiles/README.md

Per workload directory profiles/<round>_<suffix>/: kernel_stats.csv (rocprofv3 --kernel-trace --stats: average duration of
the dominant kernel), hbm_traffic.json (FETCH_SIZE / WRITE_SIZE, SQ_WAVE_CYCLES, SQ_WAIT_ANY, SQ_INSTS_* from the PMC passes:
tools/make_traffic_json.py).  frac = algorithmic bytes per launch (89 J + 10 M + 40 per env step) / the kernel's
average duration / 8 TB/s."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.make_traffic_json import WORKLOADS  # noqa: E402

LABEL = {"ta01_b65536": "headline: ta01 × 65 536, one launch per step", "ta01_b262144": "ta01 × 262 144",
         "ta01_b4096": "config 2: ta01 × 4 096", "syn15x15_b65536": "synthetic 15×15, per-env tables × 65 536",
         "ta41_b16384": "config 3: ta41 SPT × 16 384", "syn50x20_b8192": "config 4, one GPU's share: synthetic 50×20 × 8 192",
         "syn50x20_b65536": "config 4 whole on one GPU: synthetic 50×20 × 65 536", "mixed_b32768": "config 5 padded: mixed ta01–80 × 32 768",
         "mixed_bucketed_b32768": "config 5, shape classes in ONE grid: mixed ta01–80 × 32 768",
         "mixed_by_shape_b32768": "config 5 padded, envs ordered by shape class, class bodies on the padded rows"}


def kernel_avg_us(stats_csv, needle):
    needle = needle.replace(" ", "")
    for r in csv.DictReader(open(stats_csv)):
        if needle in r["Name"].replace(" ", ""):
            return float(r["AverageNs"]) / 1e3, int(r["Calls"])
    raise KeyError((stats_csv, needle))


def table(rnd):
    traffic = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    rows = ["| workload (rocprofv3 dir) | dominant kernel | launches traced | avg µs per launch | algorithmic MB per launch | frac of 8 TB/s | "
            "HBM-side traffic ÷ algorithmic | wave-cycles per env step | waiting (SQ_WAIT_ANY) | VALU / SALU per wave |", "|---|---|---|---|---|---|---|---|---|---|"]
    for key, (suffix, kern, label, _alg, _batch) in WORKLOADS.items():
        d = os.path.join(ROOT, "profiles", f"{rnd}_{suffix}")
        ent = traffic.get(key)
        if not os.path.isdir(d) or not ent or ent.get("round") != rnd:
            continue
        us, calls = kernel_avg_us(os.path.join(d, "kernel_stats.csv"), kern)
        alg = ent["algorithmic_bytes_per_launch_if_every_env_steps"]
        frac = alg / (us * 1e-6) / 8e12
        rows.append(f"| {LABEL.get(key, key)} (`{rnd}_{suffix}`) | `{label}` | {calls} | {us:.2f} | {alg / 1e6:.2f} | **{frac:.3f}** | {ent['ratio']:.3f} | "
                    f"{ent.get('wave_cycles_per_env_step', float('nan')):.0f} | {100 * ent.get('wait_fraction', float('nan')):.0f} % | "
                    f"{ent.get('valu_per_wave', float('nan')):.0f} / {ent.get('salu_per_wave', float('nan')):.0f} |")
    sub2 = os.path.join(ROOT, "profiles", f"{rnd}_ta01_sub2", "kernel_stats.csv")
    if os.path.isfile(sub2):
        us, calls = kernel_avg_us(sub2, WORKLOADS["ta01_b65536"][1])
        rows.append(f"| the headline as 2 sub-batches on 2 streams (`{rnd}_ta01_sub2`) | half-batch kernels, two in flight | {calls} | {us:.2f} | | | | | | |")
    return "\n".join(rows)


def main():
    rnd = sys.argv[1]
    text = table(rnd)
    print(text)
    if "--write" in sys.argv:
        path = os.path.join(ROOT, "profiles", "README.md")
        s = open(path).read()
        begin, end = f"<!-- {rnd}-table-begin (tools/profile_table.py {rnd} --write) -->", f"<!-- {rnd}-table-end -->"
        if begin not in s:
            raise SystemExit(f"{path} has no '{begin}' marker")
        s = re.sub(re.escape(begin) + r".*?" + re.escape(end), lambda m: begin + "\n" + text + "\n" + end, s, flags=re.S)
        open(path, "w").write(s)


if __name__ == "__main__":
    main()
