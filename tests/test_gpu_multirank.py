"""Two ranks on ONE GPU over gloo: rehearses the sharded training and bench code paths that run
with one rank per GPU over RCCL in production."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _launch(module_or_script, extra, nproc=2, timeout=300):
    env = dict(os.environ, MARL_AMD_DIST_BACKEND="gloo", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + module_or_script + extra
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_two_rank_training_keeps_replicas_identical(tmp_path):
    script = tmp_path / "run.py"
    script.write_text(
        "import os, sys, torch\n"
        "from marl_amd import train\n"
        "import marl_amd.agents.mappo as m\n"
        "keep = {}\n"
        "orig = m.MAPPO.update\n"
        "def upd(self, obs):\n"
        "    orig(self, obs); keep['p'] = torch.cat([p.detach().reshape(-1) for p in self.policy.parameters()]).cpu()\n"
        "m.MAPPO.update = upd\n"
        "train.main(sys.argv[1:])\n"
        f"torch.save(keep['p'], os.path.join({str(tmp_path)!r}, 'params%s.pt' % os.environ['RANK']))\n")
    r = _launch([str(script)], ["--num-agents", "2", "--num-envs", "8", "--layout", "overcooked_cramped_room_v0",
                                "--num-steps", "16", "--total-steps", str(8 * 2 * 16 * 3), "--ppo-epoch", "2",
                                "--max-steps", "20", "--data-path", str(tmp_path / "data"), "--algorithm", "cmappo",
                                "--graph-rollout", "on"])
    assert r.returncode == 0, r.stderr[-3000:]
    p0 = torch.load(tmp_path / "params0.pt")
    p1 = torch.load(tmp_path / "params1.pt")
    assert torch.isfinite(p0).all()
    assert torch.equal(p0, p1), "gradient averaging did not keep the two replicas in step"


def test_two_rank_bench_reports_whole_job_throughput():
    r = _launch([os.path.join(ROOT, "bench.py")], ["--gpus", "2", "--steps", "64", "--warmup", "8", "--envs", "4096"])
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 64
    # value counts both ranks' rows
    assert abs(out["value"] - 2 * 4096 * 2 * 64 / (out["ms_per_step"] * 1e-3 * 64)) / out["value"] < 1e-6
    assert "cpu_baseline" not in out
    # the policy-driven rate is reported at every rank count, again over both ranks' rows
    pd = out["policy_driven"]
    assert pd["n_gpus"] == 2
    assert abs(pd["value"] - 2 * 4096 * 2 * pd["steps"] / (pd["ms_per_step"] * 1e-3 * pd["steps"])) / pd["value"] < 1e-6


def test_bench_starts_its_own_ranks_when_typed_without_a_launcher():
    """`python bench.py --gpus 2 ...` as typed (no torch.distributed.run in front): the process turns
    into the launcher, two fresh ranks run, one JSON line comes back that says how many ranks the
    collective library saw."""
    env = dict(os.environ, MARL_AMD_DIST_BACKEND="gloo", PYTHONPATH=ROOT)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "64", "--warmup", "8",
                        "--envs", "4096"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["world_size"] == 2 and out["config"]["all_reduce_ranks"] == 2
    assert out["config"]["process_group"] == "gloo" and out["config"]["rccl_ranks"] is None
    assert out["config"]["envs_total"] == 2 * 4096
    assert len(out["roofline"]["us_per_launch_per_rank"]) == 2
    assert out["roofline"]["us_per_launch"] == pytest.approx(max(out["roofline"]["us_per_launch_per_rank"]), rel=1e-6)
    assert 0 < out["roofline"]["frac_wall"] <= out["roofline"]["frac"] * 1.001
