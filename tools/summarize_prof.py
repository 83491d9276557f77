"""Condense the rocprofv3 CSV output of tools/gpu_profile.sh (a directory holding trace/, pmc1/, pmc2/, fetch/,
write/) into a handful of small summary CSVs written next to them -- these are what gets copied into profiles/.

    python tools/summarize_prof.py gpurun_out/prof_<tag>
"""
import collections
import csv
import glob
import os
import shutil
import statistics
import sys


def short(name):
    i = name.find("jss_")
    if i < 0:
        return name[:60]
    j = name.find("(", i)
    return name[i:j if j > 0 else None]


def main(out):
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(out, "kernel_stats.csv"))
        print(open(f).read()[:2500])
    for d in ("pmc1", "pmc2", "fetch", "write"):
        files = glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(files[0])):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        with open(os.path.join(out, f"pmc_{d}_summary.csv"), "w") as fh:
            fh.write("kernel,counter,dispatches,median_per_dispatch,min,max\n")
            for k, v in acc.items():
                for c, x in v.items():
                    fh.write(f'"{k}",{c},{len(x)},{statistics.median(x):.6g},{min(x):.6g},{max(x):.6g}\n')
                    if len(x) >= 50:
                        print(f"{k:48s} {c:22s} median/dispatch {statistics.median(x):.5g}  (n={len(x)})")


if __name__ == "__main__":
    main(sys.argv[1])
