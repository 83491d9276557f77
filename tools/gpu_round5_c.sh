# Round 5, GPU session C: what profiles/r05_* came from.  The whole -m gpu suite, the driver-form and the default bench line,
# rocprofv3 kernel traces + PMC passes (tools/gpu_profile.sh) of every benchmarked workload in its one-launch-per-step form,
# trajectory mode, and two A/Bs (the fused grid's Params copied up front; trajectory kernels with arguments in place).
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05c
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
         "mixed_bucketed --launch eager --workload mixed --batch 32768 --bucketed"; do
  set -- $w; tag=$1; shift
  timeout 900 bash tools/gpu_profile.sh r05_$tag "$@" > $O/profile_$tag.log 2>&1
  tail -2 $O/profile_$tag.log
done
for i in 1 2; do
for v in shipped multicopy; do
  L=$R/jssenv_amd/libjss_hip.so; [ $v != shipped ] && L=$R/variants/$v/libjss_hip.so
  JSSENV_AMD_LIB=$L timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 5 --workload mixed --batch 32768 --bucketed --launch sub2 --detail $O/mc_${v}_k200_$i.json > /dev/null 2>&1
  JSSENV_AMD_LIB=$L timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 --workload mixed --batch 32768 --bucketed --launch sub2 --detail $O/mc_${v}_k20_$i.json > /dev/null 2>&1
done
for v in shipped allinplace; do
  L=$R/jssenv_amd/libjss_hip.so; [ $v != shipped ] && L=$R/variants/$v/libjss_hip.so
  JSSENV_AMD_LIB=$L timeout 300 python tools/gpu_traj_probe.py >> $O/traj_probe.txt 2>&1
done
done
cat $O/traj_probe.txt
timeout 400 python tools/gpu_session_halves.py > $O/session_halves.txt 2>&1; cat $O/session_halves.txt
python - <<'PY'
import glob, json, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "r05c")
for f in sorted(glob.glob(O + "/mc_*.json")):
    d = json.load(open(f))
    print(f"{os.path.basename(f):36s} {d['value']/1e9:7.3f} G  frac {d['roofline']['frac']:.3f}  us/step {d['ms_per_step']*1e3:.2f}  min {d['windows']['min']/1e9:.3f} max {d['windows']['max']/1e9:.3f}")
PY
