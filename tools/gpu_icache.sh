cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for w in "--launch eager" "--workload synthetic50x20 --batch 8192 --launch eager" "--workload mixed --batch 32768 --launch eager"; do
  rm -rf /tmp/ic; rocprofv3 -f csv --kernel-include-regex "jss_.*5, [01]" --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVES -d /tmp/ic -o ic -- python $R/bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras $w > /tmp/ic.log 2>&1
  python - <<PY
import csv, glob, collections, statistics
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/ic/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: statistics.median(v[-50:]) for k, v in acc.items()}
print("$w", {k: round(v) for k, v in m.items()}, "miss rate %.3f" % (m.get("SQC_ICACHE_MISSES", 0) / max(1, m.get("SQC_ICACHE_REQ", 1))), "ifetch/wave %.0f" % (m.get("SQ_IFETCH", 0) / max(1, m.get("SQ_WAVES", 1))))
PY
done
