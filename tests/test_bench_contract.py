"""The recorded bench line (profiles/r01_bench_n1.json, written by bench.py on the GPU box) carries every field the driver's
contract names, and the figures inside it are mutually consistent."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_recorded_bench_line_has_the_contract_fields():
    d = json.load(open(os.path.join(ROOT, "profiles", "r01_bench_n1.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "u8" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # achieved = algorithmic bytes per launch group / live HIP-event span
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    # 2 963 001 B/frame for pyramid+FAST, + 2 234 734 for the blur when it is scheduled inside the pass
    assert r["algorithmic_bytes_per_frame"] == (2963001 + 2234734 if "blur" in r["kernel"] else 2963001)
    assert r["traffic"] >= r["algorithmic_bytes_per_launch"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "frames/s"
    frames = d["config"]["frames_per_step_per_gpu"] * d["n_gpus"]
    assert abs(d["value"] - frames / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
