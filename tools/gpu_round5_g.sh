set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05g
mkdir -p $O
cd $R
for i in 1 2; do
for v in shipped nocounters; do
  L=$R/jssenv_amd/libjss_hip.so; [ $v != shipped ] && L=$R/variants/$v/libjss_hip.so
  for w in "head_e --launch eager" "head_s --launch sub2" "c4_s --workload synthetic50x20 --batch 8192 --launch sub2" "c3_s --instance ta41 --policy SPT --batch 16384 --launch sub2"; do
    set -- $w; tag=$1; shift
    JSSENV_AMD_LIB=$L timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 5 --detail $O/${v}_${tag}_$i.json "$@" > /dev/null 2>&1
  done
done
done
python - <<'PY'
import glob, json, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "r05g")
for f in sorted(glob.glob(O + "/*.json")):
    d = json.load(open(f))
    print(f"{os.path.basename(f):30s} us/step {d['ms_per_step']*1e3:.2f}  event-time us/step {d['roofline']['kernel_ms']*1e3:.2f}")
PY
