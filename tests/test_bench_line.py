"""The line bench.py ends with is what the driver parses: it must stay ONE small JSON object whatever was measured
(round 4's 28 KB line was not parsed at all).  CPU-only: compact_line() is a pure function of the result dict."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline", "configs")
ROOFLINE_REQUIRED = ("bound", "achieved", "peak", "unit", "frac", "frac_own_bytes", "traffic", "kernel", "gpu_ms_per_step_events",
                     "alg_bytes_per_env_step")


def canned(fat=False):
    """A result dict shaped like a full --extras run (fat: every string and side object blown up far beyond what a run makes)."""
    pad = "x" * (20000 if fat else 40)
    side = {"workload": "w" + pad, "batch": 4096, "policy": "random", "value": 634012345.678, "min": 1.0, "max": 2.0, "windows": 80,
            "unit": "env steps/s", "ms_per_step": 0.0063, "roofline_frac": 0.12345678, "kernel": "k" + pad, "launch": "l" + pad,
            "trajectory": {"note": pad, "value": 1.0}, "external_actions": {"steps_per_launch": {"launch": pad}},
            "step_only": {"value": 5.9e8, "roofline_frac": 0.11223344, "launch": pad, "note": pad, "windows": {"n": 40}}}
    out = {
        "metric": "env steps/sec (batched)", "value": 4269739860.024089, "unit": "env steps/s", "n_gpus": 1, "steps": 20, "warmup": 5,
        "ms_per_step": 0.015289901057258248, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32",
        "data": "ta01 (the reference's Taillard instance; no dataset involved)",
        "windows": {"n": 151, "steps_each": 20, "statistic": "median", "min": 2.9e9, "max": 4.4e9, "below_90pct_of_median": 2, "p10": 4.1e9,
                    "timed_seconds_total": 0.05},
        "launch": "2 sub-batches on 2 HIP streams per step (jss_rollout_steps)" + pad,
        "config": {"workload": "ta01 (15x15) one instance shared by the batch, random masked policy fused with step(), batch 65536 envs "
                               "per GPU, full obs/mask/reward/done written every step, auto-restart" + pad,
                   "batch_per_gpu": 65536, "global_batch": 65536, "parallelism": "env-shard x1", "policy": "random"},
        "roofline": {"bound": "hbm", "achieved": 6511.353286536735, "peak": 8000.0, "unit": "GB/s", "frac": 0.8139191608170919,
                     "frac_of_measured_peak": 1.035, "frac_own_bytes": 0.537, "achieved_own_bytes": 4300.0, "frac_note": pad * 3,
                     "wave_cycles_per_env_step": 934.0, "wait_fraction": 0.37, "valu_per_wave": 527, "salu_per_wave": 331,
                     "frac_gpu_time": 0.8336, "achieved_gpu_time": 6669.1, "measured_peak": 6290.0, "traffic": 65688657,
                     "traffic_source": pad * 5, "kernel": "jss_packed_kernel<16,kRollout1,kTabLdsC>", "gpu_ms_per_step_events": 0.014928150177,
                     "alg_bytes_per_env_step": 1525, "env_steps_per_launch": 65283.9},
        "single_launch_per_step": {"value": 3.43e9, "min": 1.0, "max": 2.0, "gpu_ms_per_step_events": 0.0184, "launch": pad, "roofline_frac": 0.65,
                                   "roofline_frac_gpu_time": 0.68},
        "episodes_finished": 5000.0, "mean_makespan": 1836.123456, "mean_reward_per_step": 0.01,
        "cpu_baseline": {"value": 24952.115284599313, "unit": "env steps/s", "cores": 1, "kind": "port",
                         "sample": "ta01, random masked policy (README.md:53-64) + step() to completion, whole episodes for 10.0 s" + pad,
                         "implementation": pad, "host": {"logical": 256}},
        "cpu_baseline_c_oracle": {"value": 2.2e7, "sample": pad}, "cpu_baseline_twin": {"value": 3.9e7, "one_core": {"sample": pad}},
        "host": {"hsa_enable_interrupt": "0"}, "csrc_sha16": "50504be1ffa90bbe",
        "step_only": dict(side), "trajectory": dict(side), "external_actions": dict(side), "facade_b1": dict(side), "batch_x4": dict(side),
    }
    for key in bench.CONFIG_KEYS:
        out[key] = dict(side)
    out["config5_mixed_bucketed_batch32768"] = {"value": None, "error": "RuntimeError: " + pad}
    return out


def check(line, out):
    assert "\n" not in line and len(line.encode()) < bench.COMPACT_MAX_BYTES, len(line)
    d = json.loads(line)
    for k in REQUIRED:
        assert k in d, k
    for k in ROOFLINE_REQUIRED:
        assert k in d["roofline"], k
    assert d["value"] == float(f"{out['value']:.7g}") and d["ms_per_step"] == float(f"{out['ms_per_step']:.7g}")
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["higher_is_better"] is True
    for k in ("workload", "batch_per_gpu", "policy"):
        assert k in d["config"], k
    for k in ("value", "cores", "kind"):
        assert k in d["cpu_baseline"], k
    assert d["cpu_baseline"]["kind"] in ("port", "reference")
    assert set(d["configs"]) == set(bench.CONFIG_KEYS.values())
    assert d["configs"]["c2_ta01_b4096_random"] == 0.1235
    assert d["configs"]["c5_mixed_b32768_bucketed"] is None               # a failed extra is a null, not a paragraph
    # what a caller of the reference's own interface gets -- jss_step with ITS actions, one launch per step -- next to every fused figure
    assert d["configs_step_only"]["headline"] == 0.1235 and d["configs_step_only"]["c4_syn50x20_b8192_one_gpu_share"] == 0.1122
    assert "syn15x15_b65536_per_env_tables" in d["configs"] and "c5_mixed_b32768_padded_interleaved" in d["configs"]
    assert d["windows"]["p10"] == 4.1e9 and d["windows"]["n"] == 151
    return d


def test_compact_line_of_a_full_run_is_small_and_complete():
    out = canned()
    d = check(bench.compact_line(out, detail_files=["bench_detail.json"]), out)
    assert d["detail"] == ["bench_detail.json"] and d["roofline"]["single_launch"]["gpu_ms_per_step_events"] == 0.0184
    assert d["configs_env_steps_per_s"]["c3_ta41_b16384_spt"] == 6.34e8


def test_compact_line_stays_small_whatever_the_run_printed():
    out = canned(fat=True)
    check(bench.compact_line(out, detail_files=["bench_detail.json", "gpurun_out/bench_detail.json"]), out)


def test_compact_line_of_a_headline_only_run():
    out = {k: v for k, v in canned().items() if not k.startswith(("config2", "config3", "config4", "config5", "single_launch", "synthetic15x15"))}
    out["cpu_baseline"] = None                                              # --no-cpu-baseline / N > 1
    line = bench.compact_line(out)
    d = json.loads(line)
    assert len(line) < bench.COMPACT_MAX_BYTES and d["cpu_baseline"] is None and d["configs"] == {} and d["value"] > 0


def test_compact_line_of_a_multi_gpu_run_carries_the_rank_spread():
    out = canned()
    out.update(n_gpus=8, ranks={"value_min": 5.1e8, "value_max": 5.4e8, "numa_pinned": True, "numa_node_rank0": 0, "cpus_rank0": 24},
               cpu_baseline=None, config4_sharded=dict(out["config2_ta01_batch4096_random"]))
    line = bench.compact_line(out)
    d = json.loads(line)
    assert len(line) < bench.COMPACT_MAX_BYTES and d["ranks"] == {"value_min": 5.1e8, "value_max": 5.4e8, "numa_pinned": True}
