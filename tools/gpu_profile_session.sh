# rocprofv3 kernel traces of the external-action forms (GPU box).  Summaries land in gpurun_out/prof_r04_<tag>/.
# Usage: bash tools/gpu_profile_session.sh         (configs 2, 3, 4: jss_steps with its counters, the session forms traced only:
# counter collection serialises kernels, and a resident kernel cannot be serialised with the post / wait kernels that feed it)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for CFG in 2 3 4; do
  for FORM in steps session lockstep; do
    OUT=$R/gpurun_out/prof_r04_${FORM}_c$CFG
    mkdir -p $OUT
    CMD="python $R/tools/session_workload.py --config $CFG --form $FORM --K 20 --windows 30"
    timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1; echo "trace $FORM c$CFG rc=$?"
    if [ $FORM = steps ]; then
      timeout 300 rocprofv3 -f csv --kernel-include-regex "jss_.*kernel" --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- $CMD > $OUT/fetch.log 2>&1; echo "fetch rc=$?"
      timeout 300 rocprofv3 -f csv --kernel-include-regex "jss_.*kernel" --pmc WRITE_SIZE -d $OUT/write -o write -- $CMD > $OUT/write.log 2>&1; echo "write rc=$?"
      timeout 300 rocprofv3 -f csv --kernel-include-regex "jss_.*kernel" --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_ANY -d $OUT/pmc1 -o pmc1 -- $CMD > $OUT/pmc1.log 2>&1; echo "pmc1 rc=$?"
    fi
    grep -h "^{" $OUT/trace.log | tail -1 > $OUT/workload.json
    (cd $R && python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1)
    find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
  done
done
du -sh $R/gpurun_out
