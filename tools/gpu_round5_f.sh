# Round 5, GPU session F (final sources): what profiles/r05_* came from.  The whole -m gpu suite, the driver-form and the default bench line,
# rocprofv3 kernel traces + PMC passes (tools/gpu_profile.sh) of every benchmarked workload in its one-launch-per-step form,
# trajectory mode, and two A/Bs (the fused grid's Params copied up front; trajectory kernels with arguments in place).
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05f
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > $O/bench_k20.out 2> $O/bench_k20.err; tail -c 3500 $O/bench_k20.out; tail -4 $O/bench_k20.err
cp bench_detail.json $O/bench_k20_detail.json
( time timeout 600 python bench.py ) > $O/bench_default.out 2> $O/bench_default.err; tail -c 3500 $O/bench_default.out
cp bench_detail.json $O/bench_default_detail.json
for w in "ta01_single --launch eager" "ta01_sub2 --launch sub2" "ta01_b4096 --launch eager --batch 4096" "ta41 --launch eager --instance ta41 --policy SPT --batch 16384" \
         "syn15x15 --launch eager --workload synthetic15x15" "syn50x20 --launch eager --workload synthetic50x20 --batch 8192" \
         "syn50x20_b65536 --launch eager --workload synthetic50x20 --batch 65536" "mixed --launch eager --workload mixed --batch 32768" \
         "mixed_bucketed --launch eager --workload mixed --batch 32768 --bucketed" "mixed_by_shape --launch eager --workload mixed --batch 32768 --by-shape"; do
  set -- $w; tag=$1; shift
  timeout 900 bash tools/gpu_profile.sh r05_$tag "$@" > $O/profile_$tag.log 2>&1
  tail -2 $O/profile_$tag.log
done
timeout 600 bash tools/gpu_profile_traj.sh r05_traj > $O/profile_traj.log 2>&1; tail -8 $O/profile_traj.log
( time timeout 900 python bench.py --steps 20 --warmup 5 --extras ) > $O/bench_k20_extras.out 2> $O/bench_k20_extras.err; tail -c 600 $O/bench_k20_extras.out; tail -4 $O/bench_k20_extras.err
cp bench_detail.json $O/bench_k20_extras_detail.json
