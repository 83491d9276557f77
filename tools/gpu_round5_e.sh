# Round 5, GPU session E: staggered workgroup starts in the one-wavefront-per-env kernel (A/B builds)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05e
mkdir -p $O
cd $R
for i in 1 2; do
for v in shipped stagger16 stagger40 stagger80; do
  L=$R/jssenv_amd/libjss_hip.so; [ $v != shipped ] && L=$R/variants/$v/libjss_hip.so
  for w in "c4e --workload synthetic50x20 --batch 8192 --launch eager" "c4s --workload synthetic50x20 --batch 8192 --launch sub2" "c4all --workload synthetic50x20 --batch 65536 --launch sub2" "c5e --workload mixed --batch 32768 --launch eager" "c5s --workload mixed --batch 32768 --launch sub2"; do
    set -- $w; tag=$1; shift
    JSSENV_AMD_LIB=$L timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 5 --detail $O/st_${v}_${tag}_$i.json "$@" > /dev/null 2>&1
  done
done
done
python - <<'PY'
import glob, json, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "r05e")
for f in sorted(glob.glob(O + "/st_*.json")):
    d = json.load(open(f))
    print(f"{os.path.basename(f):36s} {d['value']/1e9:7.3f} G  frac {d['roofline']['frac']:.3f}  us/step {d['ms_per_step']*1e3:.2f}  min {d['windows']['min']/1e9:.3f} max {d['windows']['max']/1e9:.3f}")
PY
