cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_l2b; mkdir -p $OUT
for cfg in "0 65536" "4 65536" "0 8192" "4 8192"; do
  set -- $cfg
  JSSENV_AMD_LIB=$R/variants/profiling.so rocprofv3 -f csv --kernel-include-regex "jss_packed_kernel.*5, 2" --pmc TCC_EA0_RDREQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/m$1_$2 -o m -- python $R/tools/gpu_pmc_ablate.py $1 $2 ta01 > $OUT/m$1_$2.log 2>&1
  python - <<PY
import csv, glob, collections, statistics
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/m$1_$2/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("ablate mask $1 batch $2: " + "  ".join("%s %.0f" % (k, statistics.median(v[-50:])) for k, v in sorted(acc.items())))
PY
done
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -delete
