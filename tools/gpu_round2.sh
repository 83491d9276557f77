# One GPU session: smoke, parity tests, headline bench (all extras), launch-mode A/B, occupancy A/B.
# Usage (GPU box): bash tools/gpu_round2.sh [tag]
set -x
T=${1:-a}
O=gpurun_out/r2$T
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench.log | cut -c1-6000
for m in eager graph sub2 sub4; do
  timeout 300 python bench.py --launch $m --no-cpu-baseline --no-extras > $O/bench_$m.log 2>&1; echo "$m rc=$?"; tail -1 $O/bench_$m.log | cut -c1-700
done
for lib in "" variants/occ7.so; do
  for w in "--workload synthetic50x20 --batch 8192" "--workload mixed --batch 32768" "--workload mixed --batch 32768 --bucketed"; do
    for m in eager sub2; do
      tag=$(echo "$lib $w $m" | tr -c 'a-zA-Z0-9\n' '_')
      JSSENV_AMD_LIB=${lib:+$PWD/$lib} timeout 300 python bench.py $w --launch $m --no-cpu-baseline --no-extras > $O/bench_$tag.log 2>&1
      echo "== lib=[$lib] $w $m rc=$?"; tail -1 $O/bench_$tag.log | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print('   value %.4g  ms/step %.4f  frac %.3f  %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['launch']))
except Exception as e: print('   parse failed', e)"
    done
  done
done
timeout 60 python bench.py --gpus 2 --steps 5 --warmup 1 > $O/bench_gpus2.log 2>&1; echo "gpus2 on one GPU rc=$? (must be non-zero)"; tail -2 $O/bench_gpus2.log
du -sh gpurun_out
