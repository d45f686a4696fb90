"""The committed bench line (profiles/r01_bench_n1.json, written by bench.py on the GPU box) carries every field of the
bench contract: the metric of BASELINE.json, the roofline object and the CPU baseline."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    with open(os.path.join(ROOT, "profiles", "r01_bench_n1.json")) as f:
        line = json.loads(f.read().strip().splitlines()[-1])
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    assert line["metric"] == base["metric"] and line["unit"] == "env.step/s" and line["higher_is_better"] is True
    for k in ("value", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["scaling"] == "weak" and line["vs_baseline"] is None and "workload" in line["config"]
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # achieved = launches in flight x algorithmic bytes per launch / average launch duration
    assert abs(r["achieved"] - r["launches_in_flight"] * r["bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["bytes_per_launch"] == 1688 * r["games_per_launch"] and r["traffic"] is None or r["traffic"] > r["bytes_per_launch"]
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "env.step/s" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
