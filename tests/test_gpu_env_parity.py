"""HIP step/reset/featurize vs the CPU oracle, bit for bit, through the C ABI."""
import numpy as np
import pytest
import torch

from marl_amd.layouts import get_layout

pytestmark = pytest.mark.gpu

CASES = [
    # layout, agents, feature, envs, max_steps, steps
    ("overcooked_cramped_room_v0", 2, "global_obs", 257, 60, 200),
    ("overcooked_cramped_room_v0", 2, "local_obs", 64, 1000, 120),
    ("overcooked_forced_coordination_v0", 2, "global_obs", 130, 45, 150),
    ("overcooked_counter_circuit_v0", 2, "global_obs", 100, 70, 150),
    ("overcooked_coordination_ring_v0", 4, "local_obs", 99, 50, 150),
    ("overcooked_coordination_ring_v0", 4, "global_obs", 65, 50, 150),
    ("large_overcooked_layout", 4, "global_obs", 70, 150, 320),
    ("overcooked_asymmetric_advantages_v0", 2, "Minimal_spatial", 70, 40, 100),
    ("overcooked_cramped_room_v0", 2, "Minimal_spatial_other_agent_aware", 70, 40, 100),
    ("overcooked_cramped_room_v0", 4, "only_direction", 33, 30, 80),
    ("overcooked_cramped_room_v0", 1, "global_obs", 50, 30, 80),
    # three agents: four lanes per env, one idle; 48 rows per wave
    ("overcooked_coordination_ring_v0", 3, "global_obs", 77, 40, 160),
    ("overcooked_cramped_room_v0", 3, "local_obs", 130, 35, 120),
    ("large_overcooked_layout", 3, "Minimal_spatial_other_agent_aware", 45, 60, 130),
]


def _biased_actions(rng, n):
    # uniform actions rarely finish a soup; lean on interact so pots, plates and deliveries get exercised
    return rng.choice(7, size=n, p=[0.14, 0.14, 0.14, 0.14, 0.3, 0.04, 0.1]).astype(np.int64)


def _assert_state_equal(gpu_env, ora):
    snap = gpu_env.export_state()
    np.testing.assert_array_equal(snap["agents"], ora.agents)
    np.testing.assert_array_equal(snap["pots"], ora.pots)
    np.testing.assert_array_equal(snap["items"], ora.items)
    np.testing.assert_array_equal(snap["misc"], ora.misc)


@pytest.mark.parametrize("layout,A,feature,E,max_steps,steps", CASES)
def test_rollout_parity(oracle_mod, layout, A, feature, E, max_steps, steps):
    from marl_amd.env import OBS_KEY, BatchedOvercookedVecEnv

    rows = get_layout(layout)
    seed, base = 1234, 1000
    gpu = BatchedOvercookedVecEnv(rows, A, feature, E, max_steps=max_steps, seed=seed, env_index_base=base)
    ora = oracle_mod.OracleVecEnv(rows, A, feature, max_steps=max_steps, num_envs=E, seed=seed, env_index_base=base)
    obs_g, _ = gpu.reset()
    obs_o = ora.reset()
    np.testing.assert_array_equal(obs_g[OBS_KEY].cpu().numpy(), obs_o)
    _assert_state_equal(gpu, ora)
    rng = np.random.default_rng(7)
    total_reward = 0.0
    for t in range(steps):
        act = _biased_actions(rng, E * A)
        o_g, r_g, te_g, tr_g, _ = gpu.step(torch.from_numpy(act))
        o_o, r_o, te_o, tr_o = ora.step(act)
        np.testing.assert_array_equal(r_g.cpu().numpy(), r_o, err_msg=f"rewards differ at step {t}")
        np.testing.assert_array_equal(te_g.cpu().numpy(), te_o)
        np.testing.assert_array_equal(tr_g.cpu().numpy(), tr_o, err_msg=f"truncated differs at step {t}")
        bad = np.nonzero((o_g[OBS_KEY].cpu().numpy() != o_o).any(axis=1))[0]
        assert bad.size == 0, f"obs rows {bad[:8]} differ at step {t}"
        if t % 10 == 0 or tr_o.any():
            _assert_state_equal(gpu, ora)
        total_reward += float(r_o.sum())
    _assert_state_equal(gpu, ora)
    if layout == "overcooked_cramped_room_v0" and max_steps >= 40:
        assert total_reward > 0, "rollout never earned a reward: the test is not exercising pots"


def _random_layout(rng, A):
    """An enclosed grid with random contents: at least A floor cells, at most 4 pots."""
    while True:
        H, W = int(rng.integers(4, 10)), int(rng.integers(4, 12))
        grid = [["#"] * W for _ in range(H)]
        pots = 0
        for r in range(H):
            for c in range(W):
                border = r in (0, H - 1) or c in (0, W - 1)
                ch = rng.choice(list(" CUO=@#"), p=[0.0, 0.3, 0.2, 0.15, 0.15, 0.15, 0.05] if border
                                else [0.62, 0.14, 0.05, 0.05, 0.05, 0.05, 0.04])
                if ch == "U":
                    if pots == 4:
                        ch = "C"
                    else:
                        pots += 1
                grid[r][c] = str(ch)
        rows = ["".join(row) for row in grid]
        if sum(row.count(" ") for row in rows) >= max(A, 2):
            return rows


@pytest.mark.parametrize("case", range(24))
def test_random_layouts(oracle_mod, case):
    """Random kitchens (unreachable corners, pots on the border, no pot at all, ...): the lookup
    tables the host builds for the kernel against the oracle's cell scans."""
    from marl_amd.env import OBS_KEY, BatchedOvercookedVecEnv

    rng = np.random.default_rng(1000 + case)
    A = int(rng.integers(1, 5))
    rows = _random_layout(rng, A)
    feature = ["global_obs", "local_obs", "Minimal_spatial_other_agent_aware"][case % 3]
    E, steps = 40, 90
    gpu = BatchedOvercookedVecEnv(rows, A, feature, E, max_steps=40, seed=case, dist_metric=case % 2)
    ora = oracle_mod.OracleVecEnv(rows, A, feature, max_steps=40, num_envs=E, seed=case, dist_metric=case % 2)
    np.testing.assert_array_equal(gpu.reset()[0][OBS_KEY].cpu().numpy(), ora.reset(), err_msg=str(rows))
    for t in range(steps):
        act = _biased_actions(rng, E * A)
        o_g, r_g, _, tr_g, _ = gpu.step(torch.from_numpy(act))
        o_o, r_o, _, tr_o = ora.step(act)
        bad = np.nonzero((o_g[OBS_KEY].cpu().numpy() != o_o).any(axis=1))[0]
        assert bad.size == 0, f"obs rows {bad[:8]} differ at step {t} on {rows}"
        np.testing.assert_array_equal(r_g.cpu().numpy(), r_o, err_msg=f"step {t} on {rows}")
        np.testing.assert_array_equal(tr_g.cpu().numpy(), tr_o)
    _assert_state_equal(gpu, ora)


CROWDED = [
    # a corridor and two small rooms: agents queue, contest cells, try to swap and get pushed back in chains
    (["#######", "#     #", "#######"], 3), (["#######", "#     #", "#######"], 4),
    (["#####", "#   #", "# U #", "#   #", "#####"], 4), (["####", "#  #", "#  #", "####"], 3),
    (["#OU@=#", "#    #", "#C  C#", "######"], 4),
]


@pytest.mark.parametrize("rows,A", CROWDED)
@pytest.mark.parametrize("replay", [False, True])
def test_crowded_floors(oracle_mod, rows, A, replay):
    """Three and four agents with hardly any room: the serving order, the no-swap rule and the
    retreat from occupied cells (several in a row) on nearly every step; with drawn and with
    caller-given serving orders."""
    from marl_amd.env import OBS_KEY, BatchedOvercookedVecEnv

    E, steps = 150, 120
    gpu = BatchedOvercookedVecEnv(rows, A, "local_obs", E, max_steps=50, seed=5)
    ora = oracle_mod.OracleVecEnv(rows, A, "local_obs", max_steps=50, num_envs=E, seed=5)
    np.testing.assert_array_equal(gpu.reset()[0][OBS_KEY].cpu().numpy(), ora.reset())
    rng = np.random.default_rng(A * 7 + len(rows))
    moved = 0
    for t in range(steps):
        act = rng.choice(7, size=E * A, p=[0.22, 0.22, 0.22, 0.22, 0.06, 0.0, 0.06]).astype(np.int64)
        prio = rng.integers(0, 256, size=E * A).astype(np.uint8) if replay else None
        before = ora.agents[:, :, :2].copy()
        o_g, r_g, _, tr_g, _ = gpu.step(torch.from_numpy(act), prio_override=None if prio is None else torch.from_numpy(prio))
        o_o, r_o, _, tr_o = ora.step(act, prio_override=prio)
        moved += int((ora.agents[:, :, :2] != before).any(axis=2).sum())
        bad = np.nonzero((o_g[OBS_KEY].cpu().numpy() != o_o).any(axis=1))[0]
        assert bad.size == 0, f"obs rows {bad[:8]} differ at step {t} on {rows}"
        np.testing.assert_array_equal(r_g.cpu().numpy(), r_o)
        np.testing.assert_array_equal(tr_g.cpu().numpy(), tr_o)
        _assert_state_equal(gpu, ora)
    assert moved > steps * E // 4, "agents hardly moved: the test is not exercising the collision rules"


@pytest.mark.parametrize("A", [1, 2, 3, 4])
@pytest.mark.parametrize("E", [1, 17])
def test_tiny_batches_and_constant_restarts(oracle_mod, A, E):
    """One env (a single, mostly idle wave) and a ragged 17, with episodes of 3 steps so that
    every third step restarts every env inside the kernel."""
    from marl_amd.env import OBS_KEY, BatchedOvercookedVecEnv

    rows = get_layout("overcooked_coordination_ring_v0")
    for feature in ("global_obs", "only_direction"):
        gpu = BatchedOvercookedVecEnv(rows, A, feature, E, max_steps=3, seed=21, env_index_base=5)
        ora = oracle_mod.OracleVecEnv(rows, A, feature, max_steps=3, num_envs=E, seed=21, env_index_base=5)
        np.testing.assert_array_equal(gpu.reset()[0][OBS_KEY].cpu().numpy(), ora.reset())
        rng = np.random.default_rng(E * 10 + A)
        for t in range(31):
            act = rng.integers(0, 7, size=E * A)
            o_g, r_g, _, tr_g, _ = gpu.step(torch.from_numpy(act))
            o_o, r_o, _, tr_o = ora.step(act)
            np.testing.assert_array_equal(o_g[OBS_KEY].cpu().numpy(), o_o, err_msg=f"{feature} step {t}")
            np.testing.assert_array_equal(r_g.cpu().numpy(), r_o)
            np.testing.assert_array_equal(tr_g.cpu().numpy(), tr_o)
            assert bool(tr_o.all()) == ((t + 1) % 3 == 0)
        _assert_state_equal(gpu, ora)


@pytest.mark.parametrize("act_dtype", [torch.int64, torch.int32, torch.uint8])
def test_action_dtypes_and_priority_override(oracle_mod, act_dtype):
    from marl_amd.env import OBS_KEY, BatchedOvercookedVecEnv

    rows = get_layout("overcooked_coordination_ring_v0")
    E, A = 200, 4
    gpu = BatchedOvercookedVecEnv(rows, A, "global_obs", E, max_steps=1000, seed=5)
    ora = oracle_mod.OracleVecEnv(rows, A, "global_obs", max_steps=1000, num_envs=E, seed=5)
    rng = np.random.default_rng(3)
    floor = [r * len(rows[0]) + c for r, row in enumerate(rows) for c, ch in enumerate(row) if ch in " +"]
    spawn = np.stack([rng.choice(floor, size=A, replace=False) for _ in range(E)]).astype(np.uint8)
    dirs = rng.integers(0, 4, size=(E, A)).astype(np.uint8)
    obs_g, _ = gpu.reset(spawn_override=spawn, dir_override=dirs)
    obs_o = ora.reset(spawn_override=spawn, dir_override=dirs)
    np.testing.assert_array_equal(obs_g[OBS_KEY].cpu().numpy(), obs_o)
    for t in range(60):
        act = rng.integers(0, 7, size=E * A)
        prio = np.stack([rng.permutation(A) for _ in range(E)]).astype(np.uint8)
        o_g, r_g, _, _, _ = gpu.step(torch.from_numpy(act).to(act_dtype), prio_override=prio)
        o_o, r_o, _, _ = ora.step(act, prio_override=prio)
        np.testing.assert_array_equal(o_g[OBS_KEY].cpu().numpy(), o_o, err_msg=f"step {t}")
        np.testing.assert_array_equal(r_g.cpu().numpy(), r_o)
    _assert_state_equal(gpu, ora)
    # replay mode consumes no RNG draws
    assert (gpu.export_state()["misc"][:, 1] == 0).all()


def test_reset_overrides_are_checked_on_the_host_and_kept_inside_the_tables_on_the_device():
    """Host arrays naming a wall, a shared cell or a direction above 3 are refused; the same values
    handed over as device tensors (not read back) must leave a state every table lookup survives:
    agents on floor cells, directions 0-3, empty hands."""
    from marl_amd.env import BatchedOvercookedVecEnv

    rows = get_layout("overcooked_cramped_room_v0")
    E, A, W = 70, 2, len(rows[0])
    gpu = BatchedOvercookedVecEnv(rows, A, "global_obs", E, max_steps=50, seed=2)
    cells = [r * W + c for r, row in enumerate(rows) for c, ch in enumerate(row) if ch in " +"]
    good = np.tile(np.array([cells[:2]], np.uint8), (E, 1))
    wall = good.copy(); wall[3, 1] = 0
    with pytest.raises(ValueError, match="not floor"):
        gpu.reset(spawn_override=wall)
    same = good.copy(); same[5] = (cells[0], cells[0])
    with pytest.raises(ValueError, match="one cell"):
        gpu.reset(spawn_override=same)
    with pytest.raises(ValueError, match="outside 0..3"):
        gpu.reset(dir_override=np.full((E, A), 7, np.uint8))
    gpu.reset(spawn_override=torch.from_numpy(wall).cuda(), dir_override=torch.full((E, A), 7, dtype=torch.uint8, device="cuda"))
    rng = np.random.default_rng(0)
    for _ in range(20):
        gpu.step(torch.from_numpy(rng.integers(0, 7, size=E * A)))
    snap = gpu.export_state()
    floor = np.array([[ch in " +" for ch in r] for r in rows])
    assert floor[snap["agents"][..., 0], snap["agents"][..., 1]].all()
    assert (snap["agents"][..., 2] <= 3).all() and (snap["agents"][..., 3] <= 3).all()


def test_import_state_reproduces_reference_fixture():
    """states/agent_0_dropping_onion_state.npy of the reference, rebuilt by state injection."""
    import os

    from marl_amd.env import BatchedOvercookedVecEnv

    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "agent_0_dropping_onion_state.npy"))
    rows = get_layout("overcooked_cramped_room_v0")
    gpu = BatchedOvercookedVecEnv(rows, 2, "global_obs", 1)
    gpu.reset()
    snap = gpu.export_state()
    snap["agents"][0, 0] = (2, 3, 3, 1)  # agent 0 at (2,3) facing up, holding an onion
    snap["agents"][0, 1] = (3, 3, 1, 0)  # agent 1 at (3,3) facing down
    gpu.import_state(snap)
    obs = gpu.featurize().cpu().numpy()
    np.testing.assert_array_equal(obs[0:1], gold)


@pytest.mark.parametrize("layout,A,feature,E", [
    ("overcooked_cramped_room_v0", 2, "global_obs", 65536),       # the headline workload
    ("overcooked_coordination_ring_v0", 4, "local_obs", 65536),   # four agents: the env's lanes share the step
    ("overcooked_counter_circuit_v0", 2, "global_obs", 131072),   # one GPU's share of the 1M-env job
])
def test_full_size_properties(oracle_mod, layout, A, feature, E):
    """The benchmark sizes: what the oracle cannot follow quickly is checked through invariants
    (shards reproduce the whole job; featurize(state) equals the observation step returned;
    agents of an env stand on distinct cells), and a slice of the job (first and last 32 envs)
    against the oracle step by step."""
    from marl_amd.env import OBS_KEY, BatchedOvercookedVecEnv

    rows = get_layout(layout)
    S = 32
    whole = BatchedOvercookedVecEnv(rows, A, feature, E, seed=11, env_index_base=0)
    half = BatchedOvercookedVecEnv(rows, A, feature, E // 2, seed=11, env_index_base=E // 2)
    head = oracle_mod.OracleVecEnv(rows, A, feature, num_envs=S, seed=11, env_index_base=0)
    tail = oracle_mod.OracleVecEnv(rows, A, feature, num_envs=S, seed=11, env_index_base=E - S)
    ow, _ = whole.reset()
    oh, _ = half.reset()
    assert torch.equal(ow[OBS_KEY][E // 2 * A:], oh[OBS_KEY])
    np.testing.assert_array_equal(ow[OBS_KEY][:S * A].cpu().numpy(), head.reset())
    np.testing.assert_array_equal(ow[OBS_KEY][-S * A:].cpu().numpy(), tail.reset())
    g = torch.Generator(device="cpu").manual_seed(0)
    for t in range(40):
        act = torch.randint(0, 7, (E * A,), generator=g)
        ow, rw, _, _, _ = whole.step(act)
        oh, rh, _, _, _ = half.step(act[E // 2 * A:])
        assert torch.equal(ow[OBS_KEY][E // 2 * A:], oh[OBS_KEY]), f"shard diverged at step {t}"
        assert torch.equal(rw[E // 2 * A:], rh)
        o_head, r_head, _, _ = head.step(act[:S * A].numpy())
        o_tail, r_tail, _, _ = tail.step(act[-S * A:].numpy())
        np.testing.assert_array_equal(ow[OBS_KEY][:S * A].cpu().numpy(), o_head, err_msg=f"step {t}")
        np.testing.assert_array_equal(ow[OBS_KEY][-S * A:].cpu().numpy(), o_tail, err_msg=f"step {t}")
        np.testing.assert_array_equal(rw[:S * A].cpu().numpy(), r_head)
        np.testing.assert_array_equal(rw[-S * A:].cpu().numpy(), r_tail)
    assert torch.equal(whole.featurize(), ow[OBS_KEY])
    # observations are small integers: nothing outside what int8 staging can hold
    assert ow[OBS_KEY].abs().max().item() <= 30
    assert torch.equal(ow[OBS_KEY], ow[OBS_KEY].round())
    cells = torch.from_numpy(whole.export_state()["agents"][:, :, 0] * 64 + whole.export_state()["agents"][:, :, 1])
    srt = cells.sort(dim=1).values
    assert bool((srt[:, 1:] != srt[:, :-1]).all()), "two agents of an env share a cell"


def _host_threads():
    """every core this job may use (the box exports its share as OMP_NUM_THREADS)"""
    import os

    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    return max(1, min(n, share) if share > 0 else n)


FULL_SIZE = [
    ("overcooked_cramped_room_v0", 2, "global_obs", 65536),       # the headline workload
    ("overcooked_counter_circuit_v0", 2, "global_obs", 131072),   # one GPU's share of the 1M-env job
    ("overcooked_coordination_ring_v0", 4, "local_obs", 65536),   # four agents, per-agent observations
    ("overcooked_forced_coordination_v0", 2, "global_obs", 65536),  # the centralised learner's layout ('+' spawn cells)
    ("large_overcooked_layout", 4, "global_obs", 32768),          # the reference's own layout, widest rows (D = 420)
    ("overcooked_coordination_ring_v0", 3, "global_obs", 21848),  # three agents: the quad's fourth lane idles
    ("overcooked_asymmetric_advantages_v0", 2, "Minimal_spatial_other_agent_aware", 65536),
]


@pytest.mark.parametrize("layout,A,feature,E", FULL_SIZE)
def test_full_size_every_row_against_oracle(oracle_mod, layout, A, feature, E):
    """The benchmark sizes with EVERY env followed by the oracle (all host threads): every
    observation row, reward and flag of every step, the exported state every 10 steps and at
    each restart, over 60 single steps with episodes of 25 (two in-kernel restarts of every env
    at full size), then one 32-step launch of marl_env_step_many held to the same bar."""
    from marl_amd.env import OBS_KEY, BatchedOvercookedVecEnv

    rows = get_layout(layout)
    N = E * A
    gpu = BatchedOvercookedVecEnv(rows, A, feature, E, max_steps=25, seed=77, env_index_base=3 * E)
    ora = oracle_mod.OracleVecEnv(rows, A, feature, max_steps=25, num_envs=E, seed=77, env_index_base=3 * E,
                                  nthreads=_host_threads())
    D, dev = gpu.D, gpu.device
    host = torch.empty((N, D), dtype=torch.float32, pin_memory=True)  # the oracle writes here, one upload per step
    want = torch.empty((N, D), dtype=torch.float32, device=dev)

    def rows_equal(got, what):
        want.copy_(host, non_blocking=True)
        if not torch.equal(got, want):
            bad = torch.nonzero((got != want).any(dim=1)).flatten()[:8].tolist()
            raise AssertionError(f"{what}: {len(bad)}+ observation rows differ, first {bad}")

    obs_g, _ = gpu.reset()
    host.numpy()[...] = ora.reset()
    rows_equal(obs_g[OBS_KEY], "reset")
    _assert_state_equal(gpu, ora)
    rng = np.random.default_rng(E + A)
    restarts = 0
    for t in range(60):
        act = _biased_actions(rng, N)
        o_g, r_g, te_g, tr_g, _ = gpu.step(torch.from_numpy(act))
        _, r_o, te_o, tr_o = ora.step(act, out_obs=host.numpy())
        rows_equal(o_g[OBS_KEY], f"step {t}")
        np.testing.assert_array_equal(r_g.cpu().numpy(), r_o, err_msg=f"rewards differ at step {t}")
        np.testing.assert_array_equal(te_g.cpu().numpy(), te_o)
        np.testing.assert_array_equal(tr_g.cpu().numpy(), tr_o, err_msg=f"truncated differs at step {t}")
        if tr_o.any():
            assert tr_o.all()
            restarts += 1
        if t % 10 == 0 or tr_o.any():
            _assert_state_equal(gpu, ora)
    assert restarts == 2
    _assert_state_equal(gpu, ora)

    # one multi-step launch from where the single steps left off: 32 steps, another restart inside
    K = 32
    acts = np.stack([_biased_actions(rng, N) for _ in range(K)])
    obs_k = torch.empty((K, N, D), dtype=torch.float32, device=dev)
    rew_k = torch.empty((K, N), dtype=torch.float32, device=dev)
    te_k = torch.empty((K, N), dtype=torch.bool, device=dev)
    tr_k = torch.empty_like(te_k)
    gpu.step_many_into(torch.from_numpy(acts).to(dev), obs_k, rew_k, te_k, tr_k)
    rew_h, te_h, tr_h = rew_k.cpu().numpy(), te_k.cpu().numpy(), tr_k.cpu().numpy()
    for k in range(K):
        _, r_o, te_o, tr_o = ora.step(acts[k], out_obs=host.numpy())
        rows_equal(obs_k[k], f"multi-step launch, step {k}")
        np.testing.assert_array_equal(rew_h[k], r_o, err_msg=f"multi-step launch: rewards differ at step {k}")
        np.testing.assert_array_equal(te_h[k], te_o)
        np.testing.assert_array_equal(tr_h[k], tr_o)
    assert tr_h.any()
    _assert_state_equal(gpu, ora)


def test_a_million_envs_on_one_gpu_every_row(oracle_mod):
    """The whole 1,048,576-env counter_circuit job (what eight GPUs share) on a single GPU, every env
    followed by the oracle: the largest batch the benchmarks name, observation stores in their
    large-tensor cache policy, an in-kernel restart of every env on the way."""
    from marl_amd.env import OBS_KEY, BatchedOvercookedVecEnv

    rows = get_layout("overcooked_counter_circuit_v0")
    E, A = 1 << 20, 2
    N = E * A
    gpu = BatchedOvercookedVecEnv(rows, A, "global_obs", E, max_steps=8, seed=5)
    ora = oracle_mod.OracleVecEnv(rows, A, "global_obs", max_steps=8, num_envs=E, seed=5, nthreads=_host_threads())
    host = torch.empty((N, gpu.D), dtype=torch.float32, pin_memory=True)
    want = torch.empty((N, gpu.D), dtype=torch.float32, device=gpu.device)
    got = torch.empty_like(want)
    gpu.reset(out=got)
    host.numpy()[...] = ora.reset()
    want.copy_(host)
    assert torch.equal(got, want)
    rng = np.random.default_rng(1)
    for t in range(12):
        act = _biased_actions(rng, N)
        _, r_g, _, tr_g, _ = gpu.step(torch.from_numpy(act), out=got)
        _, r_o, _, tr_o = ora.step(act, out_obs=host.numpy())
        want.copy_(host)
        assert torch.equal(got, want), f"observations differ at step {t}"
        np.testing.assert_array_equal(r_g.cpu().numpy(), r_o)
        np.testing.assert_array_equal(tr_g.cpu().numpy(), tr_o)
        assert bool(tr_o.all()) == (t == 7)
    _assert_state_equal(gpu, ora)


def test_store_policy_does_not_change_rows(monkeypatch):
    """Large observation tensors are stored with a different cache hint; force each policy on the
    same small job (ragged last row group included) and compare every output."""
    from marl_amd.env import OBS_KEY, BatchedOvercookedVecEnv

    outs = []
    for nt in ("0", "1"):
        monkeypatch.setenv("MARL_AMD_STREAM_NT", nt)
        for layout, A, feat, E in (("overcooked_cramped_room_v0", 2, "global_obs", 1000),
                                   ("overcooked_coordination_ring_v0", 4, "local_obs", 333)):
            env = BatchedOvercookedVecEnv(get_layout(layout), A, feat, E, max_steps=25, seed=3)
            g = torch.Generator(device="cpu").manual_seed(1)
            o, _ = env.reset()
            got = [o[OBS_KEY].clone()]
            for _ in range(40):
                o, r, _, tr, _ = env.step(torch.randint(0, 7, (E * A,), generator=g))
                got += [o[OBS_KEY].clone(), r.clone(), tr.clone()]
            outs.append(got)
    half = len(outs) // 2
    for a, b in zip(outs[:half], outs[half:]):
        for x, y in zip(a, b):
            assert torch.equal(x, y)


def test_single_env_dict_api_without_auto_reset(oracle_mod):
    """make_env(...) itself steps like a bare cogrid env (tests/test_load.py:74-79,104-128):
    per-agent dicts, and no restart inside step()."""
    from marl_amd.env import OBS_KEY, make_env

    env = make_env(num_agents=2, layout="overcooked_forced_coordination_v0", feature="Minimal_spatial_other_agent_aware",
                   max_steps=12)
    ora = oracle_mod.OracleVecEnv(get_layout("overcooked_forced_coordination_v0"), 2, "Minimal_spatial_other_agent_aware",
                                  max_steps=12, num_envs=1, seed=0, auto_reset=False)
    obs, info = env.reset()
    oo = ora.reset()
    assert set(obs) == {0, 1} and obs[0][OBS_KEY].shape == env.observation_spaces[0][OBS_KEY].shape
    np.testing.assert_array_equal(np.stack([obs[i][OBS_KEY] for i in range(2)]), oo)
    rng = np.random.default_rng(1)
    for t in range(15):
        act = rng.integers(0, 7, size=2)
        obs, rew, term, trunc, info = env.step({0: int(act[0]), 1: int(act[1])})
        oo, ro, teo, tro = ora.step(act)
        np.testing.assert_array_equal(np.stack([obs[i][OBS_KEY] for i in range(2)]), oo)
        assert [rew[0], rew[1]] == ro.tolist()
        done = [term[i] or trunc[i] for i in range(2)]
        assert all(done) == (t + 1 >= 12)       # stays done: nothing restarted the episode
    assert ora.misc[0, 0] == 15


def test_layouts_without_a_start_cell_per_agent_are_refused(oracle_mod):
    """forced_coordination names two spawn cells: a third agent has nowhere to start.  Library and
    oracle both refuse at creation instead of stacking agents on one cell; both also refuse a grid
    with floor on its border."""
    from marl_amd._lib import MarlAmdError
    from marl_amd.env import BatchedOvercookedVecEnv

    rows = get_layout("overcooked_forced_coordination_v0")
    with pytest.raises(MarlAmdError, match="spawn"):
        BatchedOvercookedVecEnv(rows, 3, "global_obs", 4)
    with pytest.raises(ValueError):
        oracle_mod.OracleVecEnv(rows, 3, "global_obs", num_envs=4)
    with pytest.raises(MarlAmdError):
        BatchedOvercookedVecEnv(["####", "#C #", "####"], 2, "global_obs", 4)  # one floor cell, two agents
    # floor on the border of the grid (nothing beyond it): every registered layout is walled in
    open_rows = ["  O  ", " CUC ", "     ", "=   @"]
    with pytest.raises(MarlAmdError, match="enclosed"):
        BatchedOvercookedVecEnv(open_rows, 2, "global_obs", 4)
    with pytest.raises(ValueError):
        oracle_mod.OracleVecEnv(open_rows, 2, "global_obs", num_envs=4)


def test_determinism_and_reseeding():
    from marl_amd.env import OBS_KEY, BatchedOvercookedVecEnv

    rows = get_layout("overcooked_coordination_ring_v0")
    acts = torch.randint(0, 7, (30, 300 * 4), generator=torch.Generator().manual_seed(5))

    def rollout(env):
        outs = []
        for t in range(30):
            o, r, _, _, _ = env.step(acts[t])
            outs.append((o[OBS_KEY].clone(), r.clone()))
        return outs

    a = BatchedOvercookedVecEnv(rows, 4, "local_obs", 300, max_steps=9, seed=1)
    b = BatchedOvercookedVecEnv(rows, 4, "local_obs", 300, max_steps=9, seed=1)
    c = BatchedOvercookedVecEnv(rows, 4, "local_obs", 300, max_steps=9, seed=2)
    oa, ob, oc = a.reset()[0][OBS_KEY], b.reset()[0][OBS_KEY], c.reset()[0][OBS_KEY]
    assert torch.equal(oa, ob) and not torch.equal(oa, oc)
    ra, rb = rollout(a), rollout(b)
    assert all(torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) for x, y in zip(ra, rb))
    # re-keying c with seed 1 makes it replay a's episode from the start
    oc2 = c.reset(seed=1)[0][OBS_KEY]
    assert torch.equal(oc2, oa)
    rc = rollout(c)
    assert all(torch.equal(x[0], y[0]) for x, y in zip(ra, rc))


def test_step_writes_into_caller_buffer_and_rejects_bad_ones():
    from marl_amd.env import OBS_KEY, BatchedOvercookedVecEnv

    env = BatchedOvercookedVecEnv(get_layout("overcooked_cramped_room_v0"), 2, "global_obs", 70, seed=3)
    store = torch.zeros((3, env.N, env.D), device="cuda")
    env.reset(out=store[0])
    act = torch.randint(0, 7, (env.N,))
    o, *_ = env.step(act, out=store[1])
    assert o[OBS_KEY].data_ptr() == store[1].data_ptr()
    assert store[2].abs().sum() == 0 and store[1].abs().sum() > 0
    assert torch.equal(env.featurize(), store[1])
    with pytest.raises(ValueError):
        env.step(act, out=torch.zeros((env.N, env.D + 1), device="cuda")[:, :-1])   # not contiguous
    with pytest.raises(ValueError):
        env.step(act[:-1])
    with pytest.raises(ValueError):
        env.step(act, out=torch.zeros((env.N, env.D), dtype=torch.float64, device="cuda"))


@pytest.mark.parametrize("layout,A,feature,E", [
    ("overcooked_cramped_room_v0", 2, "global_obs", 258), ("overcooked_coordination_ring_v0", 4, "local_obs", 99),
    ("overcooked_coordination_ring_v0", 3, "global_obs", 76), ("overcooked_cramped_room_v0", 1, "local_obs", 64),
    ("large_overcooked_layout", 4, "global_obs", 33), ("overcooked_cramped_room_v0", 2, "only_direction", 70),
])
def test_many_steps_in_one_launch(layout, A, feature, E):
    """marl_env_step_many against the same number of marl_env_step calls: every output of every
    step and the final state, bit for bit, with episodes ending inside the launch; then both
    carry on from the state the multi-step launch left."""
    from marl_amd.env import BatchedOvercookedVecEnv

    rows = get_layout(layout)
    K = 23
    one = BatchedOvercookedVecEnv(rows, A, feature, E, max_steps=9, seed=4, env_index_base=17)
    many = BatchedOvercookedVecEnv(rows, A, feature, E, max_steps=9, seed=4, env_index_base=17)
    N, D = one.N, one.D
    dev = one.device
    one.reset(); many.reset()
    g = torch.Generator(device="cpu").manual_seed(E)
    for rnd in range(2):
        acts = torch.randint(0, 7, (K, N), generator=g).to(dev)
        prio = torch.randint(0, 256, (K, N), generator=g, dtype=torch.uint8).to(dev) if rnd == 1 and A > 1 else None
        obs1 = torch.empty((K, N, D), device=dev); rew1 = torch.empty((K, N), device=dev)
        te1 = torch.empty((K, N), dtype=torch.bool, device=dev); tr1 = torch.empty_like(te1)
        obs2, rew2, te2, tr2 = torch.empty_like(obs1), torch.empty_like(rew1), torch.ones_like(te1), torch.empty_like(tr1)
        for k in range(K):
            one.step_into(acts[k], obs1[k], rew1[k], te1[k], tr1[k], prio_override=None if prio is None else prio[k])
        many.step_many_into(acts, obs2, rew2, te2, tr2, prio_override=prio)
        assert torch.equal(obs1, obs2) and torch.equal(rew1, rew2) and torch.equal(tr1, tr2) and not te2.any()
        assert tr1.any(), "no episode ended inside the launch"
        assert torch.equal(one.state, many.state)
    with pytest.raises(ValueError):
        many.step_many_into(acts[:, :-1], obs2, rew2, te2, tr2)


def test_two_host_threads_step_their_own_handles(oracle_mod):
    """Different env handles of one kernel family, created and stepped at the same time from two
    host threads on their own streams (the library's shared bookkeeping is locked): both match
    the oracle."""
    import threading

    from marl_amd.env import OBS_KEY, BatchedOvercookedVecEnv
    from marl_amd.layouts import get_layout

    rows = get_layout("overcooked_cramped_room_v0")
    errors = []

    def worker(seed):
        try:
            torch.cuda.set_device(0)
            stream = torch.cuda.Stream()
            rng = np.random.default_rng(seed)
            with torch.cuda.stream(stream):
                for rep in range(3):
                    E = 64 + 32 * rep
                    gpu = BatchedOvercookedVecEnv(rows, 2, "global_obs", E, max_steps=15, seed=seed + rep)
                    ora = oracle_mod.OracleVecEnv(rows, 2, "global_obs", max_steps=15, num_envs=E, seed=seed + rep)
                    o, _ = gpu.reset()
                    assert np.array_equal(o[OBS_KEY].cpu().numpy(), ora.reset())
                    for _ in range(20):
                        act = rng.integers(0, 7, size=E * 2)
                        o, r, _, tr, _ = gpu.step(torch.from_numpy(act))
                        oo, ro, _, tro = ora.step(act)
                        assert np.array_equal(o[OBS_KEY].cpu().numpy(), oo)
                        assert np.array_equal(r.cpu().numpy(), ro) and np.array_equal(tr.cpu().numpy(), tro)
        except Exception as exc:  # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=worker, args=(s,)) for s in (100, 200)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_step_counter_runs_past_16_bits_without_auto_reset(oracle_mod):
    """A bare env (no restart inside step) that the caller keeps stepping: the step counter is a
    full word, so `truncated` stays set beyond 65,536 steps instead of wrapping back to 0."""
    from marl_amd.env import OBS_KEY, BatchedOvercookedVecEnv

    rows = get_layout("overcooked_cramped_room_v0")
    E, A = 40, 2
    gpu = BatchedOvercookedVecEnv(rows, A, "global_obs", E, max_steps=1000, seed=2, auto_reset=False)
    ora = oracle_mod.OracleVecEnv(rows, A, "global_obs", max_steps=1000, num_envs=E, seed=2, auto_reset=False)
    gpu.reset(); ora.reset()
    snap = ora.snapshot()
    snap["misc"][:, 0] = 65530
    snap["misc"][::2, 0] = 990  # every other env is still inside its episode
    ora.load_snapshot(snap); gpu.import_state(snap)
    rng = np.random.default_rng(0)
    for t in range(14):
        act = rng.integers(0, 7, size=E * A)
        o_g, r_g, _, tr_g, _ = gpu.step(torch.from_numpy(act))
        o_o, r_o, _, tr_o = ora.step(act)
        np.testing.assert_array_equal(o_g[OBS_KEY].cpu().numpy(), o_o)
        np.testing.assert_array_equal(tr_g.cpu().numpy(), tr_o)
        late = tr_g.cpu().numpy().reshape(E, A)
        assert late[1::2].all() and late[::2].all() == (990 + t + 1 >= 1000)
    _assert_state_equal(gpu, ora)
    assert gpu.export_state()["misc"][1, 0] == 65530 + 14
