"""`python bench.py --gpus N` without a launcher in front: the parent must build the
torch.distributed.run command, start it as a child and never touch the GPU itself."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_rank_launch_command_shape():
    import bench

    cmd = bench.rank_launch_command(8, ["--gpus", "8", "--steps", "5", "--config", "4"], 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511"
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "8", "--steps", "5", "--config", "4"]


def test_parent_launches_children_and_touches_no_gpu(monkeypatch):
    import subprocess

    import torch

    import bench

    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, **kw):
        seen["cmd"] = cmd
        return Done()

    def boom(*a, **k):
        raise AssertionError("the launcher process touched the GPU")

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    for name in ("set_device", "is_available", "current_device", "synchronize", "init"):
        monkeypatch.setattr(torch.cuda, name, boom)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 7  # the children's exit code is the parent's
    assert "--nproc-per-node=4" in seen["cmd"] and seen["cmd"][-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]


def test_rank_count_mismatch_is_refused(monkeypatch):
    import bench

    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    with pytest.raises(SystemExit, match="must agree"):
        bench.main()


def test_more_ranks_than_gpus_is_refused_before_anything_starts(monkeypatch, capsys):
    import subprocess

    import torch

    import bench

    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: (_ for _ in ()).throw(AssertionError("ranks were started")))
    monkeypatch.delenv("MARL_AMD_DIST_BACKEND", raising=False)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 2 and "1 GPU(s) are visible" in capsys.readouterr().err
