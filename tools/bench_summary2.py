"""Compact view of one bench.py JSON line: python tools/bench_summary2.py gpurun_out/bench_k20.json"""
import json
import sys

d = json.loads([ln for ln in open(sys.argv[1]) if ln.startswith("{")][-1])


def ext(e, indent="    "):
    for k, v in (e.get("external_actions") or {}).items():
        if isinstance(v, dict):
            print(f"{indent}{k:32s} {v['value']:.3g} steps/s  {v['us_per_step']:6.2f} us/step  frac {v['roofline_frac']:.3f}  "
                  f"[{v['min']:.3g} .. {v['max']:.3g}] sets/wave {v.get('env_sets_per_wavefront')} timeouts {v.get('timeouts')}")
        else:
            print(indent, k, v)


rf = d["roofline"]
print(f"headline {d['value']:.4g} steps/s  {d['ms_per_step'] * 1e3:.2f} us/step  frac {rf['frac']:.3f}  own-bytes {rf.get('frac_own_bytes')}  "
      f"gpu-time {rf['frac_gpu_time']:.3f}  windows {d['windows']}")
print("  wave_cycles/env-step", rf.get("wave_cycles_per_env_step"), "wait_fraction", rf.get("wait_fraction"))
for k in ("single_launch_per_step", "step_only", "trajectory", "policy_then_step_two_launches", "policy_then_step_pipelined", "fused_rollout", "facade_b1"):
    v = d.get(k)
    if isinstance(v, dict):
        print(f"  {k:32s}", {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()
                             if kk in ("value", "roofline_frac", "ms_per_step", "us_per_step", "fused_rule_episode_ms", "error", "eager", "graph")})
ext(d, "  ")
for k, v in d.items():
    if isinstance(v, dict) and (k.startswith("config") or k.startswith("synth") or k.startswith("batch_")):
        if v.get("value") is None:
            print(k, v)
            continue
        so = v.get("step_only") or {}
        tr = v.get("trajectory") or {}
        pp = v.get("policy_then_step_pipelined") or {}
        print(f"{k:44s} {v['value']:.4g}  {v['ms_per_step'] * 1e3:6.2f} us/step  frac {v['roofline_frac']:.3f}   step_only {so.get('roofline_frac')}   "
              f"trajectory {tr.get('roofline_frac')}   policy->step pipelined {pp.get('roofline_frac')} (n_sub {pp.get('n_sub')})")
        ext(v)
for k in ("cpu_baseline", "cpu_baseline_port", "cpu_baseline_twin"):
    v = d.get(k) or {}
    print(k, v.get("value"), v.get("cores"), v.get("kind"))
print("cpu_baseline_per_config", {k: (round(v["value"]) if isinstance(v, dict) and "value" in v else v) for k, v in (d.get("cpu_baseline_per_config") or {}).items()})
