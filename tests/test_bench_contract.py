"""The driver's contract with ``bench.py`` and ``__graft_entry__.py`` (GPU only): one JSON line on stdout with the agreed
keys, whatever else the run prints going to stderr; ``smoke()`` runs the native path."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config"}


def _bench(*args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=280,
                       cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_bench_prints_one_json_line_with_the_agreed_keys():
    d = _bench("--gpus", "1", "--steps", "20", "--warmup", "5", "--no-extras", "--no-bf16")
    assert REQUIRED <= set(d), REQUIRED - set(d)
    assert d["metric"] == "learner updates/sec" and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5
    assert d["dtype"] == "fp32" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["config"]["global_batch"] == 256 and "LunarLanderContinuous" in d["config"]["model"]
    assert d["losses_finite"] is True
    assert abs(d["value"] - 1000.0 / d["ms_per_step"]) < 0.01 * d["value"]      # value and ms_per_step say the same thing
    assert 5_000 < d["value"] < 50_000                                          # an MI355X, the fused fp32 pipeline
    assert d["implementation"].startswith("HipCoreF32")


def test_default_bench_carries_the_other_stated_metrics():
    d = _bench()
    for key in ("per_value", "d3pg_value", "ant4096_fp32_ms_per_step", "env_steps_per_sec", "pendulum_updates_to_target",
                "bf16_value", "eager_torch_value"):
        assert key in d and d[key] is not None, key
    assert d["steps"] >= 1000 and d["env_steps_per_sec"] > 1e6


def test_graft_entry_smoke_runs_the_native_kernels():
    p = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], capture_output=True, text=True,
                       timeout=280, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "smoke ok" in p.stdout
