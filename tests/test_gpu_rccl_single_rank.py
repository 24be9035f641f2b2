"""RCCL on one GPU: a one-rank "nccl" process group (torch.distributed's nccl backend is RCCL on
ROCm) with every collective of the training path forced to run -- the flat-gradient all-reduce
between backward and the gradient clip (agents/MAPPO.py:207-208), the episode-statistics
all-reduce, CMAPPO's per-step reward-sum all-reduce -- plus bench.py through its nccl branch.
One rank cannot show scaling; this shows that the RCCL code path executes on the hardware."""
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


def _launch(script_and_args, extra_env, timeout=240):
    env = dict(os.environ, PYTHONPATH=ROOT, MARL_AMD_FORCE_COLLECTIVES="1", **extra_env)
    env.pop("MARL_AMD_DIST_BACKEND", None)  # default backend on a GPU box: nccl
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + script_and_args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("algorithm,graph_rollout", [("mappo", "on"), ("cmappo", "on"), ("cmappo", "off")])
def test_training_collectives_run_over_rccl(tmp_path, algorithm, graph_rollout):
    script = tmp_path / "run.py"
    script.write_text(
        "import json, os, sys, torch, torch.distributed as dist\n"
        "from marl_amd import train, dist_utils\n"
        "import marl_amd.agents.mappo as m\n"
        "keep = {}\n"
        "orig = m.MAPPO.update\n"
        "def upd(self, obs):\n"
        "    orig(self, obs); keep['calls'] = self._reducer.calls; keep['stats'] = dict(self.last_stats)\n"
        "m.MAPPO.update = upd\n"
        "returns, freq = train.main(sys.argv[1:])\n"
        "assert dist.is_initialized() and dist.get_backend() == 'nccl' and dist.get_world_size() == 1\n"
        "print('RESULT ' + json.dumps({'calls': keep['calls'], 'stats': keep['stats'], 'returns': returns,\n"
        "                              'all_env': freq['episode_return_all_envs']}))\n")
    E, A, T, updates = 8, 2, 16, 3
    r = _launch([str(script), "--num-agents", str(A), "--num-envs", str(E), "--layout", "overcooked_cramped_room_v0",
                 "--num-steps", str(T), "--num-minibatches", "4", "--total-steps", str(E * A * T * updates),
                 "--ppo-epoch", "2", "--max-steps", "20", "--data-path", str(tmp_path / "data"),
                 "--algorithm", algorithm, "--graph-rollout", graph_rollout], {})
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    out = json.loads(line[len("RESULT "):])
    # one gradient all-reduce per minibatch step: 3 updates x 2 epochs x 4 minibatches
    assert out["calls"] == updates * 2 * 4
    assert all(v == v and abs(v) < 1e6 for v in out["stats"].values())
    # 48 driver steps = 2 episodes of 21; their statistics went through the all-reduce
    assert len(out["returns"]) == 2 and len(out["all_env"]) == 2


def test_bench_runs_through_the_nccl_branch():
    r = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "64", "--warmup", "8", "--envs", "4096",
                 "--no-cpu-baseline"], {})
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["config"]["process_group"] == "nccl"
    # the all-reduce of a device tensor of ones went through RCCL
    assert out["config"]["world_size"] == 1 and out["config"]["rccl_ranks"] == 1
    # the policy-driven leg takes its maximum over ranks through the same process group
    assert out["policy_driven"]["n_gpus"] == 1 and out["policy_driven"]["value"] > 0
