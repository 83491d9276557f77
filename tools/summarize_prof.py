"""Condense gpurun_out/prof_* (rocprofv3 CSV output of tools/gpu_profile.sh) into profiles/<name>/."""
import collections
import csv
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(name):
    out = os.path.join(ROOT, "profiles", name)
    os.makedirs(out, exist_ok=True)
    src = os.path.join(ROOT, "gpurun_out")
    shutil.copy(os.path.join(src, "prof_trace", "trace_kernel_stats.csv"), os.path.join(out, "kernel_stats.csv"))
    for d in ("pmc1", "pmc2", "fetch", "write"):
        f = os.path.join(src, f"prof_{d}", f"{d}_counter_collection.csv")
        if not os.path.isfile(f):
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            k = k[k.index("jss_"):k.index("(", k.index("jss_"))]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        with open(os.path.join(out, f"pmc_{d}_summary.csv"), "w") as fh:
            fh.write("kernel,counter,dispatches,median_per_dispatch,min,max\n")
            for k, v in acc.items():
                for c, x in v.items():
                    fh.write(f'"{k}",{c},{len(x)},{statistics.median(x):.6g},{min(x):.6g},{max(x):.6g}\n')
                    if "4>" in k:
                        print(f"{k:32s} {c:22s} {statistics.median(x):.5g}")


if __name__ == "__main__":
    main(sys.argv[1])
