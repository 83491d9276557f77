# Round 5, GPU session B: decisions.  jss_step per launch with the kernel arguments in place / by value; config 5's fused grid
# with 1, 2, 3 parts per class against the stream-per-class form; what a pass of the headline costs without its cross-lane
# reads (wrong results, timing only); how a SIMD issues mixed scalar / vector streams.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05b
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "bucketed" > $O/pytest_bucketed.log 2>&1; tail -3 $O/pytest_bucketed.log
for i in 1 2; do
for v in shipped byvalue allinplace; do
  L=$R/jssenv_amd/libjss_hip.so; [ $v != shipped ] && L=$R/variants/$v/libjss_hip.so
  JSSENV_AMD_LIB=$L timeout 300 python tools/gpu_step_probe.py >> $O/step_probe.txt 2>&1
done
done
cat $O/step_probe.txt
for i in 1 2; do
for l in "grid --launch eager" "grid --launch sub2" "grid --launch sub3" "streams"; do
  set -- $l; tag=$1$3; tag=${tag//-/}
  timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 --workload mixed --batch 32768 --bucketed --bucketed-launch "$@" --detail $O/bk_${tag}_k20_$i.json > /dev/null 2>&1
  timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 5 --workload mixed --batch 32768 --bucketed --bucketed-launch "$@" --detail $O/bk_${tag}_k200_$i.json > /dev/null 2>&1
done
for v in shipped nogrpread; do
  L=$R/jssenv_amd/libjss_hip.so; [ $v != shipped ] && L=$R/variants/$v/libjss_hip.so
  JSSENV_AMD_LIB=$L timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 5 --launch eager --detail $O/xl_${v}_head_eager_$i.json > /dev/null 2>&1
  JSSENV_AMD_LIB=$L timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 5 --launch sub2 --detail $O/xl_${v}_head_sub2_$i.json > /dev/null 2>&1
  JSSENV_AMD_LIB=$L timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 5 --launch sub2 --instance ta41 --policy SPT --batch 16384 --detail $O/xl_${v}_c3_sub2_$i.json > /dev/null 2>&1
done
done
python - <<'PY'
import glob, json, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "r05b")
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.load(open(f))
        print(f"{os.path.basename(f):40s} {d['value']/1e9:7.3f} G  frac {d['roofline']['frac']:.3f}  gpu {d['roofline']['frac_gpu_time']:.3f}  us/step {d['ms_per_step']*1e3:.2f}  min {d['windows']['min']/1e9:.3f} max {d['windows']['max']/1e9:.3f}  {d['launch'][:60]}")
    except Exception as e:
        print(f, "ERR", e)
PY
hipcc --offload-arch=gfx950 -O2 tools/ubench_valu_peak.hip -o /tmp/ubench_valu_peak > /dev/null 2>&1 && /tmp/ubench_valu_peak > $O/valu_peak.txt 2>&1; tail -14 $O/valu_peak.txt
