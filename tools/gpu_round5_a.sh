# Round 5, GPU session A: the whole -m gpu suite, the driver-form bench line, the default line, and same-box A/Bs of the
# two round-5 changes (Params read in place vs by value; config 5's shape classes in one grid vs one stream per class).
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05a
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > $O/bench_k20.out 2> $O/bench_k20.err; tail -c 4500 $O/bench_k20.out; tail -3 $O/bench_k20.err
cp bench_detail.json $O/bench_k20_detail.json
( time timeout 600 python bench.py ) > $O/bench_default.out 2> $O/bench_default.err; tail -c 4500 $O/bench_default.out
cp bench_detail.json $O/bench_default_detail.json
# A/B 1: kernel arguments read in place (shipped) vs by value (round 4's form), same box
for i in 1 2; do
for v in inplace byvalue; do
  L=$R/jssenv_amd/libjss_hip.so; [ $v = byvalue ] && L=$R/variants/byvalue/libjss_hip.so
  for w in "c4 --workload synthetic50x20 --batch 8192" "c4all --workload synthetic50x20 --batch 65536" "c5 --workload mixed --batch 32768" "head" "c3 --instance ta41 --policy SPT --batch 16384"; do
    set -- $w; tag=$1; shift
    JSSENV_AMD_LIB=$L timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 --detail $O/ab_${v}_${tag}_k20_$i.json "$@" > /dev/null 2>&1
    JSSENV_AMD_LIB=$L timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 5 --detail $O/ab_${v}_${tag}_k200_$i.json "$@" > /dev/null 2>&1
  done
done
done
# A/B 2: config 5 without padding: one grid vs one stream per class
for i in 1 2; do
for l in grid streams; do
  timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 --workload mixed --batch 32768 --bucketed --bucketed-launch $l --detail $O/ab_bucketed_${l}_k20_$i.json > /dev/null 2>&1
  timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 5 --workload mixed --batch 32768 --bucketed --bucketed-launch $l --detail $O/ab_bucketed_${l}_k200_$i.json > /dev/null 2>&1
done
done
python - <<'PY'
import glob, json, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "r05a")
for f in sorted(glob.glob(O + "/ab_*.json")):
    try:
        d = json.load(open(f))
        print(f"{os.path.basename(f):44s} {d['value']/1e9:7.3f} G  frac {d['roofline']['frac']:.3f}  gpu-time frac {d['roofline']['frac_gpu_time']:.3f}  ms/step {d['ms_per_step']:.5f}  min {d['windows']['min']/1e9:.3f} max {d['windows']['max']/1e9:.3f}  {d['launch'][:40]}")
    except Exception as e:
        print(f, "ERR", e)
PY
hipcc --offload-arch=gfx950 -O2 tools/ubench_valu_peak.hip -o /tmp/ubench_valu_peak > /dev/null 2>&1 && /tmp/ubench_valu_peak > $O/valu_peak.txt 2>&1; cat $O/valu_peak.txt
