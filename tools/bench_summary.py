"""One line per figure of a bench.py JSON line: python tools/bench_summary.py gpurun_out/<tag>/bench.json [...]"""
import json
import sys

for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f"{f}: steps={d['steps']} headline {d['value'] / 1e9:.3f} G  frac {r['frac']:.3f}  gpu_time {r.get('frac_gpu_time', 0):.3f}  "
          f"ms/step {d['ms_per_step']:.5f}  cpu {d.get('cpu_baseline', {}).get('value')}")
    for k, v in d.items():
        if isinstance(v, dict) and "value" in v and k not in ("cpu_baseline",):
            fr = v.get("roofline_frac")
            print(f"   {k:48s} {v['value'] / 1e9:8.4f} G  frac {fr if fr is None else round(fr, 3)}")
        elif k == "facade_b1":
            print(f"   facade_b1 {v.get('us_per_step')} us/step, fused episode {v.get('fused_rule_episode_ms')} ms")
