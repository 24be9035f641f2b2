"""The line `python bench.py` prints (a reduced batch, so that it runs in seconds): every field the
measurement contract names is there, no leg reports an error, and the numbers hang together."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_line_has_the_contract_fields_and_sound_legs():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--envs", "8192", "--steps", "64", "--warmup", "8",
                        "--spinup-ms", "20"], capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 64 and out["warmup"] == 8 and out["vs_baseline"] is None
    assert out["unit"] == "agent-steps/s" and out["higher_is_better"] is True and out["scaling"] == "weak"
    N = 8192 * 2
    assert abs(out["value"] - N / (out["ms_per_step"] * 1e-3)) <= 1e-6 * out["value"]
    assert "workload" in out["config"] and "model" not in out["config"]
    roof = out["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "us_per_launch", "frac_wall", "bytes_per_agent_step"):
        assert key in roof, key
    assert roof["peak"] == 8000.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    assert abs(roof["achieved"] - N * roof["bytes_per_agent_step"] / roof["us_per_launch"] / 1e3) <= 1e-6 * roof["achieved"]
    assert roof["us_per_launch"] * 1e-3 <= out["ms_per_step"] * 1.001  # the kernel fits inside the step it is part of
    for name in ("cpu_baseline", "cpu_baseline_1t"):
        base = out[name]
        assert "error" not in base, base
        assert base["kind"] == "port" and base["value"] > 0 and base["cores"] >= 1 and base["sample"]
    assert out["cpu_baseline_1t"]["cores"] == 1
    for name in ("launch_time_us", "llc_resident", "step_many", "two_shards", "policy_driven", "policy_driven_two_shards"):
        assert name in out and "error" not in out[name], (name, out.get(name))
    assert out["two_shards"]["cached_actions"]["us_per_step"] > 0
    assert out["policy_driven"]["env_launch_us_in_loop"] > 0
    assert out["step_many"]["us_per_step"] > 0 and out["llc_resident"]["us_per_launch"] > 0
