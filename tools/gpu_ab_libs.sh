# Same-box A/B of kernel-library builds under variants/ (tools/build_instrumented.py, or copies of jssenv_amd/libjss_hip.so):
#   bash tools/gpu_ab_libs.sh <tag> <libA> <libB> [...]     -> gpurun_out/<tag>/<lib>_<workload>_<run>.json + a table
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
for i in 1 2; do for v in "$@"; do
  L=$PWD/variants/$v.so
  JSSENV_AMD_LIB=$L timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 5 > $O/${v}_head200_$i.json 2>/dev/null
  JSSENV_AMD_LIB=$L timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 > $O/${v}_head20_$i.json 2>/dev/null
  JSSENV_AMD_LIB=$L timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 5 --instance ta41 --policy SPT --batch 16384 > $O/${v}_c3_$i.json 2>/dev/null
  JSSENV_AMD_LIB=$L timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 5 --workload synthetic15x15 > $O/${v}_syn15_$i.json 2>/dev/null
  JSSENV_AMD_LIB=$L timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 5 --workload synthetic50x20 --batch 8192 > $O/${v}_c4_$i.json 2>/dev/null
  JSSENV_AMD_LIB=$L timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 5 --workload mixed --batch 32768 > $O/${v}_c5_$i.json 2>/dev/null
done; done
python - <<PY
import json, glob, collections
t = collections.defaultdict(dict)
for f in sorted(glob.glob("$O/*.json")):
    lib, wl, run = f.split("/")[-1][:-5].rsplit("_", 2)
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); t[wl].setdefault(lib, []).append(round(d["value"] / 1e9, 3))
    except Exception as e:
        t[wl].setdefault(lib, []).append("ERR")
for wl in sorted(t):
    print(wl, "  ".join(f"{lib}: {v}" for lib, v in sorted(t[wl].items())))
PY
