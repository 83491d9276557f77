# PC sampling of the benchmarked kernels (rocprofv3 beta feature; GPU box).  Where do the waves spend their cycles?
# Usage: bash tools/gpu_pcsample.sh <tag> <bench args...>   -> gpurun_out/pcs_<tag>/ (raw CSV head + per-offset histogram)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pcs_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras --launch eager $*"
for method in stochastic host_trap; do
  if [ $method = stochastic ]; then UNIT=cycles; INT=65536; else UNIT=time; INT=10; fi
  timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $method --pc-sampling-unit $UNIT --pc-sampling-interval $INT \
      --kernel-trace -f csv -d $OUT/$method -o pcs -- $CMD > $OUT/$method.log 2>&1
  echo "$method rc=$?"
  f=$(find $OUT/$method -name "*pc_sampling*.csv" | head -1)
  if [ -n "$f" ] && [ $(wc -l < "$f") -gt 100 ]; then
    head -5 "$f" > $OUT/${method}_head.csv
    python $R/tools/pcsample_hist.py "$f" $(find $OUT/$method -name "*kernel_trace.csv" | head -1) > $OUT/${method}_hist.txt 2>&1
    find $OUT/$method -name "*.csv" -size +2M -delete
    break
  fi
  tail -5 $OUT/$method.log
done
find $OUT -name "*.db" -delete
du -sh $OUT
