"""Host-side pieces of bench.py that need no GPU: the byte model of the roofline, the all-reduce cost model, the core count
of the CPU baseline, and the guard that keeps the live-counter pass from nesting profilers."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, sys.argv[:1]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def test_algorithmic_bytes_follow_the_storage_type():
    B = _bench()
    assert B.algorithmic_bytes("f32") == (32.0, 96.0, 32.0)     # 8 corners x 4 B; + 64 B of gradient RMW
    assert B.algorithmic_bytes("f16") == (16.0, 80.0, 16.0)


def test_allreduce_model_matches_the_survey_figures():
    B = _bench()
    m = B.allreduce_model(512 * 2 ** 20, 8)                      # ~6.1 ms one link, ~0.9 ms all links
    assert abs(m["one_link_ring_ms"] - 6.14) < 0.02 and abs(m["all_links_ms"] - 0.877) < 0.005
    m = B.allreduce_model(4 * 2 ** 30, 8)                        # 1024^3 f32 gradients: ~49 / ~7 ms
    assert abs(m["one_link_ring_ms"] - 49.1) < 0.2 and abs(m["all_links_ms"] - 7.02) < 0.05
    assert B.allreduce_model(512 * 2 ** 20, 1) is None and B.allreduce_model(None, 8) is None


def test_host_cores_reports_what_the_process_may_use():
    B = _bench()
    n, note = B.host_cores()
    assert 1 <= n <= len(os.sched_getaffinity(0)) and "affinity" in note


def test_live_counters_are_not_collected_under_a_profiler(monkeypatch):
    """bench.py under `rocprofv3 -- python3 bench.py` must not start a rocprofv3 of its own: the child would inherit the
    profiler's preload, which initialises the GPU before rocprofv3 exec's its target."""
    B = _bench()
    assert not B.under_profiler({})
    assert B.under_profiler({"LD_PRELOAD": "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so"})
    assert B.under_profiler({"ROCP_TOOL_LIBRARIES": "x"}) and B.under_profiler({"ROCPROF_OUTPUT_PATH": "/tmp"})
    monkeypatch.setenv("ROCP_TOOL_LIBRARIES", "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so")
    monkeypatch.setattr(B.shutil, "which", lambda name: "/bin/false")   # "rocprofv3 is there" -- it must not be started
    started = []
    monkeypatch.setattr(B.subprocess, "run", lambda *a, **k: started.append(a))
    res, note = B.measure_traffic_live([])
    assert res is None and "profiler" in note and not started


def test_valu_roofline_from_the_committed_counters():
    """roofline_valu from a per-launch counter table: the round-3 profile (profiles/r03_pmc_per_launch.json) must give the
    figures VERDICT r03 read off it -- 433 / 1 016 VALU lane-instructions per voxel-step, LDS conflict ratio 0.476 / 0.400."""
    import json
    B = _bench()
    pm = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc_per_launch.json")))
    steps = 255029417
    f = B.valu_roofline(pm, False, steps)
    b = B.valu_roofline(pm, True, steps)
    assert f["kernel"].startswith("dr::brick_flat_kernel<float, 0, false") and b["kernel"].startswith("dr::brick_flat_kernel<float, 0, true")
    assert abs(f["valu_lane_instr_per_voxel_step"] - 433) < 2 and abs(b["valu_lane_instr_per_voxel_step"] - 1016) < 3
    assert abs(f["lds_conflict_ratio"] - 0.476) < 0.002 and abs(b["lds_conflict_ratio"] - 0.400) < 0.002
    assert 0.8 < f["issue_slot_utilisation"] <= 1.0 and 0.8 < b["issue_slot_utilisation"] <= 1.0
    assert 4.0 < f["waves_per_simd"] <= 5.0 and 3.0 < b["waves_per_simd"] <= 4.0
    assert B.traffic_bytes(pm, False) == int((2 * pm[f["kernel"]]["FETCH_SIZE"] + pm[f["kernel"]]["WRITE_SIZE"]) * 1024)
    assert B.valu_roofline({}, False, steps) is None


def test_roofline_bound_is_named_from_the_counters():
    """`roofline.bound` says what the SQ counters show, not what the byte model prices: on the committed
    round-4 counters the forward is VALU-issue + LDS bound, the backward VALU-issue + LDS-atomics; HBM only if the memory side is
    the busiest unit."""
    import json
    B = _bench()
    d = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_line_headline.json")))
    assert B.observed_bound(d["roofline_valu_fwd"], d["roofline_fwd"]["hbm_frac_measured"]) == "valu-issue+lds"
    assert B.observed_bound(d["roofline_valu_bwd"], d["roofline_bwd"]["hbm_frac_measured"]) == "valu-issue+lds-atomics"
    idle = dict(d["roofline_valu_fwd"], issue_slot_utilisation=0.3, lds_array_busy=0.1)
    assert B.observed_bound(idle, 0.85) == "hbm" and B.observed_bound(idle, 0.2).startswith("latency")


def test_roofline_entry_prices_evaluated_samples_and_refuses_fractions_above_one():
    """a bench line must not claim work its kernels skipped. `frac` is priced with the samples the brick
    kernels EVALUATED (one untimed counting step), says so in `frac_kind`, and a fraction above 1 is refused, not printed;
    `bound` is null until counters of the same run name it (it used to assert a profile of another configuration)."""
    B = _bench()
    # the headline: every marched sample is evaluated (a few single-sample rays go through the per-ray fallback)
    e = B.roofline_entry("march_bwd", 6.4, 96.0, 2.55e8, {"marched": 255029417, "march_fwd": 255029319, "march_bwd": 255029319}, "march_bwd")
    assert e["bound"] is None and e["model_bound"] == "hbm"
    assert abs(e["frac"] - 2.55e8 * 96 / 6.4e-3 / 1e9 / 8000.0) < 1e-3 and "EVALUATED" in e["frac_kind"] and "work-skipping" not in e["frac_kind"]
    # CT-like scene, d_volume only: 0.34 ms for 9.5e7 marched samples of which 3 % were evaluated (r05 printed frac = 8.93 here)
    e = B.roofline_entry("march_bwd", 0.34, 96.0, 9.5e7, {"marched": 95000000, "march_fwd": 40000000, "march_bwd": 2850000}, "march_bwd")
    assert e["frac"] is not None and e["frac"] < 0.2 and "work-skipping active: 0.030" in e["frac_kind"]
    assert e["evaluated_voxel_steps_per_launch"] == int(9.5e7 * 0.03) and e["voxel_steps_per_launch"] == int(9.5e7)
    # no counting step (baseline kernels): the marched count, and the guard
    e = B.roofline_entry("march_bwd", 0.34, 96.0, 2.55e8, None, "march_bwd")
    assert e["frac"] is None and e["achieved"] is None and "> 1" in e["frac_refused"] and "marched" in e["frac_kind"]
