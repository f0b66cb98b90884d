"""The ONE line bench.py prints for the driver: compact, strict JSON, the contract's keys + roofline + cpu_baseline (VERDICT r4 item 1:
round 4's single 25 KB line came back as `parsed: null`). No GPU: compact_line / kernel_roofline on a synthetic detail record."""
import json
import math
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _detail():
    kr_w = bench.kernel_roofline(16_000_000 * 20 * bench.B_WATER_STEP, 80.0, 6000, 48000)
    kr_d = bench.kernel_roofline(45_000_000 * 20 * bench.B_WIND_STEP, 2800.0, 10880, 10880)
    phases = {"particles_water": dict(kr_w, kernel="k_relax_step<water>", algorithmic_bytes_per_step=192),
              "particles_wind": dict(kr_d, kernel="k_batch_epoch<wind> (a long explanation that must not reach the compact line)", algorithmic_bytes_per_step=256),
              "grid_classify": None,
              "map_frequency": {"kernel": "k_map_frequency (freq r/w + track r, 12 B/cell)", "achieved": 5570.0, "unit": "GB/s", "frac": 0.696, "avg_launch_us": 36.0,
                                "algorithmic_bytes_per_launch": 2.0e8}}
    roof = {"bound": "hbm", "kernel": "k_batch_epoch<wind> (the relaxed engine runs wind on the exclusive schedule by default)", "achieved": kr_d["achieved"], "peak": 8000.0,
            "unit": "GB/s", "frac": kr_d["frac"], "traffic": 9.7e6, "traffic_source": "profiles/r04_pmc.json " + "x" * 400, "avg_launch_ms": 0.26, "launches": 10880,
            "launches_timed": 10880, "algorithmic_bytes_per_launch": 2.1e7, "phases": phases,
            "whole_tick": {"algorithmic_bytes_per_tick": 1.6e10, "achieved": 17.6, "unit": "GB/s", "frac": 0.0022}, "note": "n" * 2000,
            "time_share_of_tick": {"k_relax_floods": 0.31, "k_batch_epoch<wind>": 0.15, "k_grid_tiles_flow": 0.18, "k_relax_tail": 0.06, "k_relax_step<water>": 0.03}}
    phases["floods"] = dict(bench.kernel_roofline(3.0e6 * 20 * 192 + 2.1e5 * 20 * bench.B_FLOOD, 900.0, 6900, 48000), kernel="k_relax_floods", ms_per_tick=310.0)
    gate = {"pass": False, "beyond_3": [], "between_2_and_3": ["sumh", "nested"]}
    cpu = {"value": 2.5e6, "unit": "particle-steps/s", "cores": 1, "kind": "reference", "sample": "s" * 900, "ms_per_tick": 25570.9, "host_cpu": "EPYC", "host_cores_total": 256,
           "all_cores": {"value": 3.4e7, "unit": "…", "cores": 32, "note": "n" * 300}, "as_is": {"value": 1.0e6}}
    pref = {"workload": "1024x1024 ...", "controls": 8, "outliers": ["sumh"], "figures": {k: {"z": 0.5 * i - 1.0, "engine": 1.0} for i, k in enumerate(
        ("steps_water_top", "steps_wind", "nested", "sections", "standing", "water_volume", "sumh", "height_rmse"))}, "gate": gate,
            "headline_size_record": {"file": "profiles/r06_p2_reference_4096.json", "variant": "k8", "outliers": [], "gate": {"pass": True, "beyond_3": [], "between_2_and_3": []}}}
    return {"metric": "particle-steps/s", "value": 6.77e7, "unit": "particle-steps/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 917.08, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "4096x4096 layermap, ...", "engine": "relaxed", "water_generations": 4, "parallelism": "single", "parity": "P2 ...", "scaling_note": None,
                       "bit_exact_on_this_workload": {"value": 1.9e6}},
            "value_top_level": 6.3e7, "phase_ms_per_tick": {"water": 511.5, "grid": 263.4, "wind": 141.6, "freq": 0.06},
            "per_tick": [{"tick": t, "ms": 900.0 + t} for t in range(6, 26)], "steps_per_tick": {"water": 1.6e7, "wind": 4.5e7, "nested_particles": 1.3e6, "floods": 2.1e5},
            "engine_stats": {"epochs": 48000}, "digest": {"sumh": 9004586.4, "nsec": 33972926, "typehash": "41c8c5802c77fd2f", "rand_calls": 4000000},
            "roofline": roof, "cpu_baseline": cpu, "parity_vs_exact": {"big": "b" * 5000}, "parity_vs_reference": pref,
            "parity_gate": {"verdict": "fail", "rule": "no figure at |z| >= 3, at most one in [2, 3)", "this_run_1024": gate,
                            "headline_size_record": {"file": "profiles/r06_p2_reference_4096.json", "variant": "k8", "gate": {"pass": True, "beyond_3": [], "between_2_and_3": []}, "outliers": []}},
            "lines": [{"name": f"line{i}", "value": 1.0e6 * i, "ms_per_tick": 10.0 * i, "digest": {"x": "y" * 300}} for i in range(1, 9)], "lbm_wind": {"roofline": {}},
            "work_dropped": {"children_lost": 0, "pool_overflow": 0}}


def test_the_driver_line_is_compact_strict_json_with_roofline_and_cpu_baseline():
    line = json.dumps(bench.compact_line(_detail(), "/tmp/bench_detail_n1.json"), default=float, separators=(",", ":"), allow_nan=False)
    assert len(line) < 8000 and "\n" not in line
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["config"]["workload"] and "model" not in d["config"] and d["vs_baseline"] is None and d["dtype"] == "f64"
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "launches"):
        assert k in r, k
    assert r["kernel"] == "k_batch_epoch<wind>" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 * r["frac"]   # (the line rounds to six digits)
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "reference" and len(c["sample"]) <= 300
    assert d["parity_vs_reference"]["outliers"] == ["sumh"] and len(d["parity_vs_reference"]["z"]) == 8
    assert "per_tick" not in d and "parity_vs_exact" not in d and d["detail_file"].endswith(".json")
    # round 6: the gate's verdict is a top-level key (a run whose own check breaks the rule says so), the trend of the window is on the line,
    # and the launch families' shares of the tick travel with the roofline
    assert d["parity_gate"]["verdict"] == "fail" and d["parity_gate"]["this_run_1024"]["between_2_and_3"] == ["sumh", "nested"]
    assert d["ms_first_tick"] == 906.0 and d["ms_last_tick"] == 925.0
    assert abs(r["time_share_of_tick"]["k_relax_floods"] - 0.31) < 1e-9 and r["phases"]["floods"]["kernel"] == "k_relax_floods"


def test_the_headline_size_parity_record_is_read_from_its_file():
    """bench.py quotes the 4096^2 record of tools/p2_reference.py for the configuration it times by READING the committed file (VERDICT r5 weak 8:
    no literals): the default configuration (eight water generations) and the two-strip schedule are in it, inside on all eight figures."""
    rec = bench.headline_parity_record(8)
    assert rec is not None and rec["file"].startswith("profiles/r") and rec["variant"] == "k8"
    assert rec["outliers"] == [] and rec["gate"]["pass"] and len(rec["z"]) == 8 and rec["controls"] >= 32
    two = bench.headline_parity_record(8, strips=2)
    assert two is not None and two["variant"] == "k8s2" and two["gate"]["pass"]
    four = bench.headline_parity_record(4)                        # round 5's default: in the record too, and outside
    assert four is not None and not four["gate"]["pass"] and "nested" in four["outliers"]
    assert bench.headline_parity_record(3) is None


def test_the_parity_record_of_a_variant_is_the_newest_file_that_holds_it():
    """`parity_gate` names the record it read: the default configuration's comes from the primary record, the staggered schedule's (k16e140, and its two-strip
    form) from the records that measured it -- and a configuration nobody measured has none (bench.py then prints `unmeasured`, never `pass`)."""
    import bench
    import importlib.util
    spec = importlib.util.spec_from_file_location("p2_reference", os.path.join(ROOT, "tools", "p2_reference.py"))
    p2 = importlib.util.module_from_spec(spec); spec.loader.exec_module(p2)
    parse_variant = p2.parse_variant
    a, b, c = bench.headline_parity_record(8), bench.headline_parity_record(16, 1, 140), bench.headline_parity_record(16, 2, 140)
    assert a["file"] == "profiles/r06_p2_reference_4096.json" and a["variant"] == "k8" and a["gate"]["pass"] and a["outliers"] == []
    assert b["variant"] == "k16e140" and "stagger" in b["file"] and b["gate"]["pass"]
    assert c["variant"] == "k16e140s2" and c["gate"]["pass"]
    assert bench.headline_parity_record(4)["gate"]["pass"] is False                       # round 5's default: measured, outside
    assert bench.headline_parity_record(12, 1, 7) is None                                # nobody measured that
    assert parse_variant("k16e140s2r3") == {"name": "k16e140s2r3", "generations": 16, "stagger": 140, "strips": 2, "stream": 3}
    assert parse_variant("k8") == {"name": "k8", "generations": 8, "stagger": 0, "strips": 1, "stream": 0}
    with pytest.raises(ValueError):
        parse_variant("e140k16")


def test_roofline_is_bytes_per_launch_over_the_average_bracketed_launch():
    """the relaxed water step kernel is bracketed every 8th launch; the bytes of ALL launches must be spread over ALL launches."""
    steps, timed, total = 16_000_000, 6000, 48000
    kr = bench.kernel_roofline(steps * bench.B_WATER_STEP, 80.0, timed, total)
    avg_s = 80.0e-3 / timed
    assert math.isclose(kr["achieved"], steps * bench.B_WATER_STEP / total / avg_s / 1e9, rel_tol=1e-12)
    assert kr["launches"] == total and kr["launches_timed"] == timed
    same = bench.kernel_roofline(steps * bench.B_WATER_STEP, 80.0 * 8, total, total)             # every launch bracketed: the same answer
    assert math.isclose(same["achieved"], kr["achieved"], rel_tol=1e-12)
    assert bench.kernel_roofline(1.0, 0.0, 0, 0) is None


def test_nan_and_inf_do_not_reach_the_line():
    d = _detail()
    d["roofline"]["traffic"] = float("nan")
    d["value_top_level"] = float("inf")
    line = json.dumps(bench.compact_line(d, None), default=float, separators=(",", ":"), allow_nan=False)
    out = json.loads(line)
    assert out["roofline"]["traffic"] is None and out["value_top_level"] is None
