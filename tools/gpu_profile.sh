# rocprofv3 passes of one bench command (GPU box).  Summaries land in gpurun_out/prof_<tag>/.
# Usage: bash tools/gpu_profile.sh <tag> <bench args...>      e.g.  bash tools/gpu_profile.sh ta01_eager --launch eager
set -x
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras $*"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1; echo "trace rc=$?"
rocprofv3 -f csv --kernel-include-regex jss_ --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_ANY -d $OUT/pmc1 -o pmc1 -- $CMD > $OUT/pmc1.log 2>&1; echo "pmc1 rc=$?"
rocprofv3 -f csv --kernel-include-regex jss_ --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU -d $OUT/pmc2 -o pmc2 -- $CMD > $OUT/pmc2.log 2>&1; echo "pmc2 rc=$?"
rocprofv3 -f csv --kernel-include-regex jss_ --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- $CMD > $OUT/fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 -f csv --kernel-include-regex jss_ --pmc WRITE_SIZE -d $OUT/write -o write -- $CMD > $OUT/write.log 2>&1; echo "write rc=$?"
cd $R
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
# keep only the summaries (gpurun_out is capped at 64 MiB)
find $OUT -name "*.db" -delete
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*agent_info.csv" -delete
du -sh $OUT
