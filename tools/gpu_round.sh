set -x
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -5 gpurun_out/smoke.log
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 200 --warmup 20 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -3 gpurun_out/bench.log
