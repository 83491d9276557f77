# The GPU session behind profiles/r02_* (MI355X box; ~4 GPU-minutes): smoke, parity tests, the bench line,
# rocprofv3 summaries of every benchmarked kernel, launch-pipeline probe, phase ablations, copy calibration.
# Usage: bash tools/gpu_round2.sh   (instrumented build first: python tools/build_instrumented.py profiling)
set -x
O=gpurun_out/r2_final
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 900 python -m pytest tests -m gpu -q -x --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20.json 2>/dev/null
bash tools/gpu_profile.sh ta01_single --launch eager > /dev/null 2>&1
bash tools/gpu_profile.sh ta01_sub2 --launch sub2 > /dev/null 2>&1
bash tools/gpu_profile.sh syn15x15 --workload synthetic15x15 --launch eager > /dev/null 2>&1
bash tools/gpu_profile.sh ta41 --instance ta41 --policy SPT --batch 16384 --launch eager > /dev/null 2>&1
bash tools/gpu_profile.sh syn50x20 --workload synthetic50x20 --batch 8192 --launch eager > /dev/null 2>&1
bash tools/gpu_profile.sh mixed --workload mixed --batch 32768 --launch eager > /dev/null 2>&1
for w in "65536 ta01" "8192 synthetic50x20" "65536 synthetic15x15" "16384 ta41"; do echo "== $w"; JSS_NSUB=1,2,3,4 python tools/gpu_pipeline_probe.py $w 2>&1 | grep n_sub; done > $O/pipeline_probe.txt
bash tools/gpu_pmc_ablate.sh 2>&1 | grep "^mask" > $O/pmc_ablate_ta01.txt
for m in 0 31 1 4 16; do echo "== mask $m"; JSS_NSUB=1,2 JSS_ABLATE=$m JSSENV_AMD_LIB=$PWD/variants/profiling.so python tools/gpu_pipeline_probe.py 65536 ta01 2>&1 | grep n_sub; done > $O/ablate_timing_ta01.txt
python tools/gpu_copy_bw.py > $O/copy_bw.txt 2>&1
timeout 60 python bench.py --gpus 2 --steps 5 --warmup 1 > $O/bench_gpus2_on_one_gpu.log 2>&1; echo "--gpus 2 on a 1-GPU box rc=$? (must be non-zero)"
du -sh gpurun_out
