"""bench.py end to end on the GPU, short: the one-GPU line and the per-step code of an N > 1 rank (record packing on the
result stream, gather on its own stream - the gather itself is the identity without a process group)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--no-cpu-baseline", *extra]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("extra", [(), ("--no-pipeline",), ("--force-gather-path",), ("--force-gather-path", "--no-pipeline")])
def test_bench_line(extra):
    d = _run(*extra)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["scaling"] == "weak" and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["value"] > 1e6 and 0.3 < d["roofline"]["frac"] < 1.0 and d["roofline"]["bound"] == "hbm"
    assert 0.8 < d["scenes_fully_planned_frac"] < 0.95
    assert d["config"]["batches_in_flight"] == (1 if "--no-pipeline" in extra else 2)
