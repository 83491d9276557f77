# rocprofv3 passes of the bench command (GPU box).  Summaries land in gpurun_out/prof_*.
set -x
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
CMD="python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_trace -o trace -- $CMD > $OUT/prof_trace.log 2>&1; echo "trace rc=$?"
rocprofv3 -f csv --kernel-include-regex jss_ --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_ANY -d $OUT/prof_pmc1 -o pmc1 -- $CMD > $OUT/prof_pmc1.log 2>&1; echo "pmc1 rc=$?"
rocprofv3 -f csv --kernel-include-regex jss_ --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU -d $OUT/prof_pmc2 -o pmc2 -- $CMD > $OUT/prof_pmc2.log 2>&1; echo "pmc2 rc=$?"
rocprofv3 -f csv --kernel-include-regex jss_ --pmc FETCH_SIZE -d $OUT/prof_fetch -o fetch -- $CMD > $OUT/prof_fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 -f csv --kernel-include-regex jss_ --pmc WRITE_SIZE -d $OUT/prof_write -o write -- $CMD > $OUT/prof_write.log 2>&1; echo "write rc=$?"
find $OUT -name "*.csv" | xargs ls -la | head -30
tail -3 $OUT/prof_trace.log $OUT/prof_pmc1.log
python - <<'PY'
import csv, glob, os, collections
out=os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out'
for f in sorted(glob.glob(out+'/prof_*/**/*kernel_stats.csv', recursive=True)):
    print('==', f); print(open(f).read()[:3000])
for f in sorted(glob.glob(out+'/prof_*/**/*counter_collection.csv', recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:60]; acc[k][r['Counter_Name']]+=float(r['Counter_Value']); n[(k,r['Counter_Name'])]+=1
    print('==', f)
    for k,v in acc.items():
        for c,x in v.items(): print(f'{k:60s} {c:24s} total={x:.4g} per_dispatch={x/max(1,n[(k,c)]):.4g} dispatches={n[(k,c)]}')
PY

# keep only the summaries (gpurun_out is capped at 64 MiB)
mkdir -p $OUT/prof_summary
find $OUT/prof_* -name "*kernel_stats.csv" -exec cp {} $OUT/prof_summary/ \;
find $OUT/prof_* -name "*domain_stats.csv" -exec cp {} $OUT/prof_summary/ \;
find $OUT -name "*.db" -delete
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
du -sh $OUT
