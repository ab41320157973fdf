"""bench.py's result line on a GPU: one JSON object with the fields the driver reads, the dominant family bracketed in the
timed region, every family in the instrumented pass and in the leave-one-out pass."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "5", "--warmup", "1", "--no-block-calib",
                        "--no-cpu-baseline", "--no-families", *flags], capture_output=True, text=True, timeout=300, cwd=REPO)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_headline_workload_line():
    d = _run()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "families", "families_timing"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 1 and d["unit"] == "GB/s" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["family"] == "rowquant_bwd" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.3 < r["frac"] < 0.8
    # achieved = algorithmic bytes per launch / average launch duration
    assert abs(r["bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9 - r["achieved"]) < 0.01 * r["achieved"]
    fam = d["families"]
    assert set(fam) == {"rowquant_fwd", "rowquant_bwd", "act_calib_quant_fwd", "act_bwd", "gather_fwd", "gather_bwd"}
    # the families bracketed one by one cannot take less than the timed step, which brackets one of them
    assert sum(v["ms_per_step"] for v in fam.values()) >= 0.9 * d["ms_per_step"]
    for k, v in fam.items():
        assert v["ms_per_step"] > 0 and 0 < v["frac_hbm_peak"] < 0.8, (k, v)
        assert isinstance(v["ms_left_out"], float), (k, v)  # (a difference of two short sequences: noisy at 5 steps)
    # value = algorithmic bytes of the step / step time, below the copy ceiling of the GPU
    assert 0.3 * 8000 < d["value"] < 0.8 * 8000
    assert abs(d["frac_hbm_peak"] - d["value"] / 8000.0) < 1e-3


def test_row_sharded_workload_line():
    d = _run("--workload", "c4")
    assert d["scaling"] == "strong" and d["n_gpus"] == 1
    assert set(d["families"]) == {"mix_soft_fwd", "mix_soft_bwd", "mix_hard_fwd", "mix_hard_bwd", "mix2_fwd", "mix2_bwd"}
    assert d["roofline"]["family"] in d["families"] and 0.3 < d["roofline"]["frac"] < 0.8
