# One GPU round: smoke, parity tests, headline bench, the other BASELINE workloads.
set -x
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log
timeout 300 python bench.py --launch graph --no-cpu-baseline --no-extras > gpurun_out/bench_graph.log 2>&1; tail -1 gpurun_out/bench_graph.log
timeout 300 python bench.py --instance ta41 --policy SPT --batch 16384 --no-cpu-baseline --no-extras > gpurun_out/bench_ta41_spt.log 2>&1; tail -1 gpurun_out/bench_ta41_spt.log
timeout 300 python bench.py --workload synthetic50x20 --batch 8192 --no-cpu-baseline --no-extras > gpurun_out/bench_syn50x20.log 2>&1; tail -1 gpurun_out/bench_syn50x20.log
timeout 300 python bench.py --workload mixed --batch 32768 --no-cpu-baseline --no-extras > gpurun_out/bench_mixed.log 2>&1; tail -1 gpurun_out/bench_mixed.log
