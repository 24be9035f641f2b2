"""Host-side helpers of the rollout driver and the bench that need no GPU."""
import os


def test_aligned_runs_cover_the_segment_with_aligned_powers_of_two():
    from marl_amd.rollout import aligned_runs

    seen = set()
    for T in (12, 128, 200):
        for first in range(T):
            for count in range(1, T - first + 1):
                runs = aligned_runs(first, count)
                pos = first
                for start, length in runs:
                    assert start == pos and length & (length - 1) == 0 and start % length == 0
                    pos += length
                assert pos == first + count
                assert len(runs) <= 2 * max(count.bit_length(), 1)
                if T == 128:
                    seen.update(runs)
    assert len(seen) < 2 * 128  # a bounded set of graphs whatever the episode length
    assert aligned_runs(0, 128) == [(0, 128)] and aligned_runs(5, 0) == []


def test_bench_counts_the_cores_this_process_may_use(monkeypatch):
    import bench

    monkeypatch.setenv("OMP_NUM_THREADS", "3")
    threads, total, affinity, share = bench.host_threads()
    assert share == 3 and threads == min(3, total, affinity) and total == (os.cpu_count() or 1)
    monkeypatch.delenv("OMP_NUM_THREADS")
    threads, total, affinity, share = bench.host_threads()
    assert share == 0 and threads == min(total, affinity)
    assert bench.algorithmic_bytes_per_agent_step(202, 5, 2, 8) == 842.0


def test_a_failing_bench_leg_reports_its_error_and_spares_the_line(capsys):
    import bench

    out = {"value": 1.0}
    bench.guarded(out, "good", lambda: {"value": 2.0})
    bench.guarded(out, "skipped", lambda: None)
    bench.guarded(out, "bad", lambda: 1 / 0)
    assert out["good"] == {"value": 2.0} and "skipped" not in out
    assert out["bad"]["error"].startswith("ZeroDivisionError") and out["value"] == 1.0
    assert "ZeroDivisionError" in capsys.readouterr().err  # the traceback still goes to stderr


def test_scalar_log_keeps_scalars_and_writes_them_on_close(tmp_path):
    """Without tensorboard the metrics of agent_environment.py:90-94 / agents/MAPPO.py:211-218 still reach the disk."""
    import csv

    from marl_amd.rollout import ScalarLog

    log = ScalarLog(str(tmp_path / "runs" / "x"))
    log.add_scalar("episode_rewards", 1.5, 0)
    log.add_scalar("episode_rewards", 2.25, 1)
    log.add_scalar("losses/value_loss", 0.1, 7)
    assert log.scalars["episode_rewards"] == [(0, 1.5), (1, 2.25)]
    log.close()
    rows = list(csv.reader(open(tmp_path / "runs" / "x" / "scalars.csv")))
    assert rows[0] == ["tag", "step", "value"]
    assert rows[1:] == [["episode_rewards", "0", "1.5"], ["episode_rewards", "1", "2.25"], ["losses/value_loss", "7", "0.1"]]
    ScalarLog(None).close()  # nothing to write, nowhere to write it: no error
