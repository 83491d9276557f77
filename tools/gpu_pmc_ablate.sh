# Usage: [JSS_KERNEL_REGEX='jss_kernel.*5, 1'] bash tools/gpu_pmc_ablate.sh [batch] [instance|synthetic50x20]
# Per-phase dynamic instruction counts of the benchmarked kernel (GPU box): SQ_INSTS_VALU / SALU per wave with
# phases ablated (instrumented build variants/profiling.so; results of ablated runs are wrong by construction).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_abl
mkdir -p $OUT
for mask in 0 1 2 4 8 16 31; do
  JSSENV_AMD_LIB=$R/variants/profiling.so rocprofv3 -f csv --kernel-include-regex "${JSS_KERNEL_REGEX:-jss_packed_kernel.*5, 0}" --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS -d $OUT/m$mask -o m -- python $R/tools/gpu_pmc_ablate.py $mask "$@" > $OUT/m$mask.log 2>&1
  python - <<PY
import csv, glob, collections, statistics
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/m$mask/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
w = statistics.median(acc["SQ_WAVES"]) if acc["SQ_WAVES"] else 1
print("mask %2d: " % $mask + "  ".join("%s/wave %.1f" % (k[8:], statistics.median(v[-50:]) / w) for k, v in sorted(acc.items()) if k != "SQ_WAVES"))
PY
done
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -delete
