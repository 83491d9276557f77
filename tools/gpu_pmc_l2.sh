# L2 (TCC) request / hit / miss / fabric-request counters of the headline kernel, one launch per step (GPU box).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_l2; mkdir -p $OUT
CMD="python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras --launch eager"
rocprofv3 -f csv --kernel-include-regex "jss_packed_kernel.*5, 2" --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum -d $OUT/a -o a -- $CMD > $OUT/a.log 2>&1; echo "a rc=$?"
rocprofv3 -f csv --kernel-include-regex "jss_packed_kernel.*5, 2" --pmc TCC_WRITE_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_WRITEBACK_sum -d $OUT/b -o b -- $CMD > $OUT/b.log 2>&1; echo "b rc=$?"
cd $R
python - <<PY
import csv, glob, collections, statistics
for tag in ("a", "b"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(k, "median/dispatch", statistics.median(v), "n", len(v))
PY
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -delete
