# rocprofv3 kernel trace + PMC passes (tools/gpu_profile.sh) of every benchmarked workload's one-launch-per-step form and of
# the headline's pipelined form: gpurun_out/prof_<tag>/ -> copy the summaries to profiles/<round>_<tag>/ (tools/README.md).
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { tag=$1; shift; bash tools/gpu_profile.sh $tag "$@" > gpurun_out/prof_$tag.log 2>&1; tail -3 gpurun_out/prof_$tag.log; }
mkdir -p gpurun_out
run ta01_single --launch eager
run ta01_sub2 --launch sub2
run ta01_b4096 --launch eager --batch 4096
run ta41 --launch eager --instance ta41 --policy SPT --batch 16384
run syn50x20 --launch eager --workload synthetic50x20 --batch 8192
run syn50x20_b65536 --launch eager --workload synthetic50x20 --batch 65536
run mixed --launch eager --workload mixed --batch 32768 --interleaved
run mixed_by_shape --launch eager --workload mixed --batch 32768
run mixed_bucketed --workload mixed --batch 32768 --bucketed
run syn15x15 --launch eager --workload synthetic15x15
du -sh gpurun_out
