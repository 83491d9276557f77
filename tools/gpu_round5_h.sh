# Round 5, GPU session H: the padded batch ordered by shape class (class bodies on the padded rows) -- parity and A/B
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05h
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "by_shape or bucketed or multi_entry" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for i in 1 2; do
for w in "padded --workload mixed --batch 32768" "byshape --workload mixed --batch 32768 --by-shape" "bucketed --workload mixed --batch 32768 --bucketed"; do
  set -- $w; tag=$1; shift
  timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 5 --detail $O/${tag}_k200_$i.json "$@" > /dev/null 2>&1
  timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 --detail $O/${tag}_k20_$i.json "$@" > /dev/null 2>&1
  timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 5 --launch eager --detail $O/${tag}_eager_$i.json "$@" > /dev/null 2>&1
done
done
python - <<'PY'
import glob, json, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "r05h")
for f in sorted(glob.glob(O + "/*.json")):
    d = json.load(open(f))
    print(f"{os.path.basename(f):28s} {d['value']/1e9:7.3f} G  frac {d['roofline']['frac']:.3f}  us/step {d['ms_per_step']*1e3:.2f}  min {d['windows']['min']/1e9:.3f} max {d['windows']['max']/1e9:.3f}  {d['launch'][:50]}")
PY
