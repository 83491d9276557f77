# rocprofv3 passes of trajectory mode (tools/gpu_traj_probe.py: jss_trajectory, 32 steps per launch, all six workloads)
# -> gpurun_out/prof_<tag>/ like tools/gpu_profile.sh.   Usage: bash tools/gpu_profile_traj.sh r03_traj
set -x
TAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/gpu_traj_probe.py"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1; echo "trace rc=$?"
rocprofv3 -f csv --kernel-include-regex "jss_.*, 6, " --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_ANY -d $OUT/pmc1 -o pmc1 -- $CMD > $OUT/pmc1.log 2>&1; echo "pmc1 rc=$?"
rocprofv3 -f csv --kernel-include-regex "jss_.*, 6, " --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- $CMD > $OUT/fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 -f csv --kernel-include-regex "jss_.*, 6, " --pmc WRITE_SIZE -d $OUT/write -o write -- $CMD > $OUT/write.log 2>&1; echo "write rc=$?"
cd $R
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1; grep ", 6, " $OUT/summary.txt
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
grep "traj K" $OUT/trace.log
