# One GPU session of round 4 (MI355X box): smoke, parity tests, the bench line (default and the driver's short form), the
# facade probe, rocprofv3 summaries of the benchmarked kernels and of the external-action forms (jss_steps, step session).
# Usage: bash tools/gpu_round4.sh <tag> [profile workloads...]   e.g. bash tools/gpu_round4.sh r04 ta01_single ta01_sub2 b4096 ta41 syn50x20 mixed syn15x15 session
set -x
TAG=${1:-r04}; shift
O=gpurun_out/$TAG
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 1200 python -m pytest tests -m gpu -q --timeout 1200 -W ignore > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -3 $O/bench_default.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2>/dev/null; echo "bench20 rc=$?"
timeout 200 python tools/gpu_facade_probe.py > $O/facade_probe.txt 2>&1
for w in "$@"; do
  case $w in
    ta01_single) bash tools/gpu_profile.sh ${TAG}_ta01_single --launch eager > /dev/null 2>&1 ;;
    ta01_sub2) bash tools/gpu_profile.sh ${TAG}_ta01_sub2 --launch sub2 > /dev/null 2>&1 ;;
    syn15x15) bash tools/gpu_profile.sh ${TAG}_syn15x15 --workload synthetic15x15 --launch eager > /dev/null 2>&1 ;;
    ta41) bash tools/gpu_profile.sh ${TAG}_ta41 --instance ta41 --policy SPT --batch 16384 --launch eager > /dev/null 2>&1 ;;
    syn50x20) bash tools/gpu_profile.sh ${TAG}_syn50x20 --workload synthetic50x20 --batch 8192 --launch eager > /dev/null 2>&1 ;;
    mixed) bash tools/gpu_profile.sh ${TAG}_mixed --workload mixed --batch 32768 --launch eager > /dev/null 2>&1 ;;
    batch_x4) bash tools/gpu_profile.sh ${TAG}_ta01_b262144 --batch 262144 --launch eager > /dev/null 2>&1 ;;
    b4096) bash tools/gpu_profile.sh ${TAG}_ta01_b4096 --batch 4096 --launch eager > /dev/null 2>&1 ;;
    session) bash tools/gpu_profile_session.sh > $O/profile_session.log 2>&1 ;;
  esac
done
du -sh gpurun_out
