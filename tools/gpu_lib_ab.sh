# Same-box A/B of library builds on bench.py workloads.  Usage (GPU box, through gpurun):
#   bash tools/gpu_lib_ab.sh <tag> "<lib1> <lib2> ..." [reps]     lib = "shipped" or a file under variants/ (built by
#   tools/build_instrumented.py, shipped with the tree);  results: gpurun_out/<tag>/<lib>_<workload>_<rep>.json + summary.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; LIBS=$2; REPS=${3:-2}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for rep in $(seq 1 $REPS); do
for lib in $LIBS; do
  if [ "$lib" = shipped ]; then unset JSSENV_AMD_LIB; else export JSSENV_AMD_LIB=$R/variants/$lib.so; fi
  for w in "head_s20 --launch sub2 --steps 20 --warmup 5" "head_s200 --launch sub2 --steps 200 --warmup 10" "head_e200 --launch eager --steps 200 --warmup 10" \
           "c3_s200 --instance ta41 --policy SPT --batch 16384 --launch sub2 --steps 200 --warmup 10" \
           "syn15_s200 --workload synthetic15x15 --launch sub2 --steps 200 --warmup 10" \
           "c4_s200 --workload synthetic50x20 --batch 8192 --launch sub2 --steps 200 --warmup 10" \
           "c5_s200 --workload mixed --batch 32768 --launch sub2 --steps 200 --warmup 10"; do
    set -- $w; name=$1; shift
    timeout 300 python bench.py --no-extras --no-cpu-baseline --detail $O/${lib}_${name}_$rep.json "$@" > /dev/null 2>&1
  done
done
done
unset JSSENV_AMD_LIB
python - <<PY | tee $O/summary.txt
import glob, json, os, collections
rows = collections.defaultdict(list)
for f in sorted(glob.glob("$O/*.json")):
    lib, rest = os.path.basename(f)[:-5].split("_", 1)
    name = rest.rsplit("_", 1)[0]
    d = json.load(open(f))
    rows[name].append((lib, d["ms_per_step"] * 1e3, d["roofline"]["frac"], d["roofline"]["gpu_ms_per_step_events"] * 1e3))
for name, rs in rows.items():
    print(f"== {name} ==")
    for lib, us, frac, ev in rs:
        print(f"  {lib:16s} {us:7.2f} us/step (wall)  frac {frac:.3f}   {ev:7.2f} us/step (HIP events)")
PY
