"""The recorded bench lines (profiles/r02_bench_*.json, written by bench.py on the GPU box) carry every field the driver's
contract names, and the figures inside them are mutually consistent."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALG = {"c2": (2963001, 2234734), "c5": (17023027, 12838642)}       # pyramid+FAST, blur (bytes per frame)


@pytest.mark.parametrize("cfg", ["c2", "c5", "c3", "c4"])
def test_recorded_bench_line_has_the_contract_fields(cfg):
    d = json.load(open(os.path.join(ROOT, "profiles", "r02_bench_%s.json" % cfg)))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "u8" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # achieved = algorithmic bytes per launch group / live span of those launches
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    if cfg in ALG:                                                   # the blur is scheduled inside the pass
        assert r["algorithmic_bytes_per_frame"] == sum(ALG[cfg]) and "blur" in r["kernel"]
        assert r["traffic"] is not None and 1.0 <= r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.3
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["unit"] == "frames/s" and c["cores"] == (1 if cfg in ("c2", "c5") else 2)
    frames = d["config"]["frames_per_step_per_gpu"] * d["n_gpus"]
    assert abs(d["value"] - frames / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    # value counts the results-to-host copy; the device-resident figure can only be higher, and the host must not be the bound
    assert d["value_device_resident"] >= 0.98 * d["value"] and d["host_copy_matches_device"] is True
    assert d["host_enqueue_ms_per_step"] < 0.5 * d["ms_per_step"]
    assert abs(d["gpu_wall_ms_per_step"] - d["ms_per_step"]) < 0.1 * d["ms_per_step"]        # wall within 10 % of the GPU-side time
    if cfg == "c3":
        assert d["stereo_and_triangulation_match_oracle"] is True
