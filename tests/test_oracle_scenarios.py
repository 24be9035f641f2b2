"""Hand-computed known-answer tests of the Overcooked dynamics (the reference ships none:
its tests assert tensor shapes only).  Each scenario injects a state into the CPU oracle,
steps it with chosen actions and compares against values worked out by hand from the rules
in oracle/overcooked_oracle.h.  The GPU suite then holds the HIP kernels to the oracle.

cramped_room:        col 0123456
              row 0      #######
              row 1      #CCUCC#
              row 2      #O   O#
              row 3      #C   C#
              row 4      #C=C@C#
              row 5      #######
"""
import numpy as np
import pytest

from marl_amd.layouts import get_layout

UP, DOWN, LEFT, RIGHT, INTERACT, TOGGLE, NOOP = range(7)
D_RIGHT, D_DOWN, D_LEFT, D_UP = range(4)
NONE, ONION, PLATE, SOUP = range(4)
W = 7


def make(oracle_mod, agents, A=2, max_steps=1000, layout="overcooked_cramped_room_v0", feature="global_obs"):
    env = oracle_mod.OracleVecEnv(get_layout(layout), A, feature, max_steps=max_steps, num_envs=1)
    env.reset(spawn_override=np.array([[r * env.W + c for (r, c, _, _) in agents]], np.uint8),
              dir_override=np.array([[d for (_, _, d, _) in agents]], np.uint8))
    for a, (r, c, d, inv) in enumerate(agents):
        env.agents[0, a] = (r, c, d, inv)
    return env


def step(env, acts, prio=None):
    obs, rew, term, trunc = env.step(np.array(acts), prio_override=None if prio is None else np.array([prio], np.uint8))
    return obs, rew, term, trunc


def test_move_turns_and_moves(oracle_mod):
    env = make(oracle_mod, [(2, 2, D_UP, NONE), (3, 4, D_UP, NONE)])
    step(env, [RIGHT, NOOP])
    assert tuple(env.agents[0, 0]) == (2, 3, D_RIGHT, NONE)
    step(env, [DOWN, NOOP])
    assert tuple(env.agents[0, 0]) == (3, 3, D_DOWN, NONE)


def test_walking_into_counter_turns_only(oracle_mod):
    env = make(oracle_mod, [(2, 1 + 1, D_DOWN, NONE), (3, 4, D_UP, NONE)])
    step(env, [UP, NOOP])  # (1,2) is a counter
    assert tuple(env.agents[0, 0]) == (2, 2, D_UP, NONE)
    assert env.misc[0, 0] == 1


def test_contested_cell_goes_to_higher_priority(oracle_mod):
    for prio, winner in (([0, 1], 0), ([1, 0], 1)):
        env = make(oracle_mod, [(2, 2, D_UP, NONE), (2, 4, D_UP, NONE)])
        step(env, [RIGHT, LEFT], prio=prio)  # both want (2,3)
        pos = [tuple(env.agents[0, a, :2]) for a in range(2)]
        assert pos[winner] == (2, 3)
        assert pos[1 - winner] == [(2, 2), (2, 4)][1 - winner]
        # the loser still turned
        assert env.agents[0, 0, 2] == D_RIGHT and env.agents[0, 1, 2] == D_LEFT


def test_swap_is_blocked_and_following_is_allowed(oracle_mod):
    env = make(oracle_mod, [(2, 2, D_UP, NONE), (2, 3, D_UP, NONE)])
    step(env, [RIGHT, LEFT], prio=[0, 1])
    assert tuple(env.agents[0, 0, :2]) == (2, 2) and tuple(env.agents[0, 1, :2]) == (2, 3)
    step(env, [RIGHT, RIGHT], prio=[0, 1])  # agent 0 follows into the cell agent 1 vacates
    assert tuple(env.agents[0, 0, :2]) == (2, 3) and tuple(env.agents[0, 1, :2]) == (2, 4)


def test_standing_agent_blocks(oracle_mod):
    env = make(oracle_mod, [(2, 2, D_UP, NONE), (2, 3, D_UP, NONE)])
    step(env, [RIGHT, NOOP])
    assert tuple(env.agents[0, 0, :2]) == (2, 2)


def test_three_agents_refused_mover_keeps_its_cell(oracle_mod):
    # a2 (first) takes a1's cell; a0 (second) takes (2,3); a1 (last) wanted (2,3) too, is refused
    # and stays -- so a2, who moved onto a1's cell, has to go back
    env = make(oracle_mod, [(2, 2, D_UP, NONE), (3, 3, D_UP, NONE), (3, 4, D_UP, NONE)], A=3)
    step(env, [RIGHT, UP, LEFT], prio=[1, 2, 0])
    assert [tuple(env.agents[0, a, :2]) for a in range(3)] == [(2, 3), (3, 3), (3, 4)]


def test_onion_stack_pot_cook_plate_deliver(oracle_mod):
    env = make(oracle_mod, [(2, 1 + 0, D_LEFT, NONE), (3, 3, D_DOWN, NONE)])
    env.agents[0, 0] = (2, 1 + 1, D_LEFT, NONE)  # (2,2); the onion stack is two cells to the left
    total = np.zeros(2)

    def go(a0, a1=NOOP):
        _, r, _, _ = step(env, [a0, a1])
        total[:] += r
        return r

    go(LEFT)                                   # to (2,1)? no: (2,1) is the onion stack -> turn only
    assert tuple(env.agents[0, 0]) == (2, 2, D_LEFT, NONE)
    go(LEFT)
    assert tuple(env.agents[0, 0, :2]) == (2, 2)
    for k in range(3):
        # face the stack, take an onion
        go(LEFT)
        r = go(INTERACT)
        assert env.agents[0, 0, 3] == ONION and r[0] == 0
        go(RIGHT)                              # to (2,3), below the pot at (1,3)
        go(UP)                                 # turn to the pot (blocked move)
        assert tuple(env.agents[0, 0, :3]) == (2, 3, D_UP)
        r = go(INTERACT)                       # onion into the pot
        assert r[0] == np.float32(0.1) and r[1] == 0
        assert env.agents[0, 0, 3] == NONE and env.pots[0, 0, 0] == k + 1
        go(LEFT)                               # back to (2,2)
    assert tuple(env.pots[0, 0]) == (3, 30 - 1)  # the tick of the step after the third onion already ran
    # fetch a plate: stack at (4,2)
    go(DOWN)                                   # to (3,2)
    go(DOWN)                                   # (4,2) is the plate stack: turn only
    go(INTERACT)
    assert env.agents[0, 0, 3] == PLATE
    go(UP); go(RIGHT); go(UP)                  # (2,2) -> (2,3) -> face pot
    assert tuple(env.agents[0, 0, :3]) == (2, 3, D_UP)
    # cooking is not done yet: interact does nothing and pays nothing
    timer_before = env.pots[0, 0, 1]
    r = go(INTERACT)
    assert r[0] == 0 and env.agents[0, 0, 3] == PLATE and env.pots[0, 0, 1] == timer_before - 1
    while env.pots[0, 0, 1] > 1:
        go(NOOP)
    # the step whose tick brings the timer to 0 can already plate, but the reward looks at
    # the state BEFORE the step (timer 1): no plating reward
    snap = env.snapshot()
    r = go(INTERACT)
    assert env.agents[0, 0, 3] == SOUP and r[0] == 0 and tuple(env.pots[0, 0]) == (0, 30)
    # same situation, waiting one more step first: reward 0.3
    env.load_snapshot(snap)
    go(NOOP)
    assert tuple(env.pots[0, 0]) == (3, 0)
    r = go(INTERACT)
    assert env.agents[0, 0, 3] == SOUP and r[0] == np.float32(0.3) and r[1] == 0
    # deliver at (4,4): walk (2,3)->(2,4)->(3,4), face down
    go(RIGHT); go(DOWN); go(DOWN)
    assert tuple(env.agents[0, 0, :3]) == (3, 4, D_DOWN)
    r = go(INTERACT)
    assert r[0] == 1.0 and r[1] == 1.0          # shared
    assert env.agents[0, 0, 3] == NONE
    assert env.items.sum() == 0


def test_both_agents_feed_a_pot_with_one_slot_left(oracle_mod):
    # pot (1,3) holds 2 onions; agent 0 below it, agent 1 cannot reach it in cramped room,
    # so use coordination_ring where two cells touch a pot: pots at (1,4) and (2,5);
    # cell (2,4) touches both. Instead test the reward/placement order on one pot with the
    # second agent facing the same pot from... only one floor cell touches the cramped pot, so
    # check the full-pot case: reward is not paid and the onion stays in hand.
    env = make(oracle_mod, [(2, 3, D_UP, ONION), (3, 3, D_DOWN, NONE)])
    env.pots[0, 0] = (3, 12)
    _, r, _, _ = step(env, [INTERACT, NOOP])
    assert r[0] == 0 and env.agents[0, 0, 3] == ONION and tuple(env.pots[0, 0]) == (3, 11)


def test_drop_on_floor_blocks_and_pickup(oracle_mod):
    env = make(oracle_mod, [(2, 2, D_RIGHT, ONION), (3, 4, D_UP, NONE)])
    step(env, [INTERACT, NOOP])                 # drop on the empty floor cell (2,3)
    assert env.agents[0, 0, 3] == NONE and env.items[0, 2 * W + 3] == ONION
    obs, _, _, _ = step(env, [RIGHT, NOOP])     # the onion blocks the way
    assert tuple(env.agents[0, 0, :2]) == (2, 2)
    can_move = obs[0, 97:101]                   # [right, left, down, up]
    assert can_move.tolist() == [0, 0, 1, 0]    # right: onion, left: stack, down: floor, up: counter
    # ClosestObj(Onion): one loose onion at (2,3): (agent - object) = (0, -1)
    assert obs[0, 27:35].tolist() == [0, -1, 0, 0, 0, 0, 0, 0]
    step(env, [INTERACT, NOOP])                 # pick it up again
    assert env.agents[0, 0, 3] == ONION and env.items.sum() == 0


def test_cannot_drop_onto_an_agent(oracle_mod):
    env = make(oracle_mod, [(2, 2, D_RIGHT, ONION), (2, 3, D_UP, NONE)])
    step(env, [INTERACT, NOOP])
    assert env.agents[0, 0, 3] == ONION and env.items.sum() == 0


def test_counter_place_and_take(oracle_mod):
    env = make(oracle_mod, [(2, 2, D_UP, PLATE), (3, 4, D_UP, NONE)])
    step(env, [INTERACT, NOOP])                 # (1,2) is a counter
    assert env.agents[0, 0, 3] == NONE and env.items[0, 1 * W + 2] == PLATE
    obs = env.featurize()
    assert obs[0, 35:37].tolist() == [1, 0]     # ClosestObj(Plate): (2-1, 2-2)
    step(env, [INTERACT, NOOP])
    assert env.agents[0, 0, 3] == PLATE and env.items.sum() == 0
    # a full hand cannot take from a stack
    env.agents[0, 0] = (2, 2, D_LEFT, PLATE)
    step(env, [INTERACT, NOOP])
    assert env.agents[0, 0, 3] == PLATE


def test_toggle_and_noop_do_nothing(oracle_mod):
    env = make(oracle_mod, [(2, 3, D_UP, ONION), (3, 3, D_DOWN, NONE)])
    before = env.snapshot()
    _, r, _, _ = step(env, [TOGGLE, NOOP])
    assert (r == 0).all()
    after = env.snapshot()
    np.testing.assert_array_equal(before["agents"], after["agents"])
    np.testing.assert_array_equal(before["pots"], after["pots"])
    assert after["misc"][0, 0] == before["misc"][0, 0] + 1


def test_truncation_and_auto_reset(oracle_mod):
    env = make(oracle_mod, [(2, 3, D_UP, ONION), (3, 3, D_DOWN, NONE)], max_steps=3)
    env.items[0, 1 * W + 1] = SOUP
    for t in range(2):
        _, _, term, trunc = step(env, [NOOP, NOOP])
        assert not term.any() and not trunc.any()
    obs, _, term, trunc = step(env, [NOOP, NOOP])
    assert trunc.all() and not term.any()
    # the env restarted inside the call: clock, kitchen and hands are fresh, and the
    # observation handed back is the new episode's first one
    assert env.misc[0, 0] == 0 and env.items.sum() == 0 and (env.agents[0, :, 3] == NONE).all()
    np.testing.assert_array_equal(obs, env.featurize())


def test_pot_features_and_next_to_pot_encoding(oracle_mod):
    env = make(oracle_mod, [(2, 3, D_UP, NONE), (3, 3, D_DOWN, NONE)])
    expect = {  # pots (count, timer) -> (NextToPot row, status index, timer feature)
        (0, 30): (0, 1, -1), (2, 30): (1, 0, -1), (3, 17): (2, 2, 17), (3, 0): (3, 3, 0),
    }
    for (cnt, tim), (row, sidx, tfeat) in expect.items():
        env.pots[0, 0] = (cnt, tim)
        o = env.featurize()[0]
        ntp = np.zeros(16); ntp[row * 4 + 3] = 1          # the pot is above: adjacency index 3
        assert o[11:27].tolist() == ntp.tolist()
        pot = o[71:82]
        status = np.zeros(4); status[sidx] = 1
        assert pot[0] == 1 and pot[1:5].tolist() == status.tolist()
        assert pot[5] == cnt and pot[6] == tfeat
        assert pot[7:].tolist() == [1, 0, 1, 3]
        assert (o[82:93] == 0).all()                      # the layout has a single pot
        # the agent one cell further away sees the same pot, not adjacent
        o1 = env.featurize()[1]
        assert (o1[11:27] == 0).all() and o1[71 + 7:71 + 11].tolist() == [2, 0, 1, 3]


def test_spawn_points_are_respected(oracle_mod):
    rows = get_layout("overcooked_forced_coordination_v0")
    spawn = {r * len(rows[0]) + c for r, row in enumerate(rows) for c, ch in enumerate(row) if ch == "+"}
    env = oracle_mod.OracleVecEnv(rows, 2, "global_obs", num_envs=200, seed=4)
    env.reset()
    cells = env.agents[..., 0] * env.W + env.agents[..., 1]
    assert set(np.unique(cells)) == spawn
    assert (cells[:, 0] != cells[:, 1]).all()
    # both orders occur
    assert len({tuple(c) for c in cells}) == 2


def test_random_spawn_is_uniform_over_floor(oracle_mod):
    rows = get_layout("overcooked_cramped_room_v0")
    env = oracle_mod.OracleVecEnv(rows, 2, "global_obs", num_envs=6000, seed=1)
    env.reset()
    cells = env.agents[..., 0] * env.W + env.agents[..., 1]
    assert (cells[:, 0] != cells[:, 1]).all()
    counts = np.bincount(cells.reshape(-1), minlength=42)
    floor = [r * 7 + c for r in (2, 3) for c in (2, 3, 4)]
    assert counts.sum() == counts[floor].sum()
    assert counts[floor].min() > 0.8 * 2000 and counts[floor].max() < 1.2 * 2000
    dirs = np.bincount(env.agents[..., 2].reshape(-1), minlength=4)
    assert dirs.min() > 0.8 * 3000


@pytest.mark.parametrize("metric", [0, 1])
def test_distance_metric_changes_only_the_order(oracle_mod, metric):
    rows = get_layout("overcooked_counter_circuit_v0")
    env = oracle_mod.OracleVecEnv(rows, 2, "global_obs", num_envs=1, dist_metric=metric)
    env.reset(spawn_override=np.array([[2 * 10 + 2, 4 * 10 + 7]], np.uint8), dir_override=np.zeros((1, 2), np.uint8))
    o = env.featurize()[0]
    counters = o[63:71].reshape(4, 2)
    d = np.abs(counters).sum(1) if metric == 0 else (counters ** 2).sum(1)
    assert (np.diff(d) >= 0).all()


def test_reward_event_bookkeeping_by_hand(oracle_mod):
    """agent_environment.py:66-76 on a hand-made step: env 0 pots an onion and plates, env 1
    delivers (both agents get 1.0, one of them 1.1), env 2 idles."""
    r = np.array([0.1, 0.3, 1.0, np.float32(0.1 + 1.0), 0.0, 0.0], dtype=np.float32)
    totals, first = oracle_mod.reward_events(r, 3, 2)
    assert totals.tolist() == [1 + 3 + 10 + 11, 1, 1, 1]
    assert first.tolist() == [float(np.float32(0.1) + np.float32(0.3)) / 2, 0.0, 1.0, 1.0]
    # four agents: the reference still looks at rows 0..1 only
    r4 = np.array([0.0, 0.0, 1.0, 1.0], dtype=np.float32)
    totals, first = oracle_mod.reward_events(r4, 1, 4)
    assert totals.tolist() == [20, 1, 0, 0] and first.tolist() == [0.0, 0.0, 0.0, 0.0]


def test_convoy_of_four_moves_as_one(oracle_mod):
    """Four agents in a row on an open stretch of the large layout, all stepping right: each
    follows into the cell the one ahead vacates, whatever order they are served in."""
    rows = get_layout("large_overcooked_layout")
    r = next(i for i, row in enumerate(rows) if "     " in row)
    c0 = rows[r].index("     ")
    for prio in ([0, 1, 2, 3], [3, 2, 1, 0], [2, 0, 3, 1]):
        env = make(oracle_mod, [(r, c0 + k, D_UP, NONE) for k in range(4)], A=4, layout="large_overcooked_layout")
        step(env, [RIGHT] * 4, prio=prio)
        assert [tuple(env.agents[0, a, :2]) for a in range(4)] == [(r, c0 + k + 1) for k in range(4)], prio
        assert all(env.agents[0, a, 2] == D_RIGHT for a in range(4))


def test_plate_at_a_pot_that_is_not_ready_does_nothing(oracle_mod):
    env = make(oracle_mod, [(2, 3, D_UP, PLATE), (3, 1 + 1, D_UP, NONE)])  # agent 0 faces the pot at (1,3)
    for pot in ((0, 30), (2, 30), (3, 7)):  # empty, filling, cooking
        env.pots[0, 0] = pot
        _, rew, _, _ = step(env, [INTERACT, NOOP])
        assert env.agents[0, 0, 3] == PLATE and rew.tolist() == [0.0, 0.0]
    env.pots[0, 0] = (3, 0)  # done
    _, rew, _, _ = step(env, [INTERACT, NOOP])
    assert env.agents[0, 0, 3] == SOUP and tuple(env.pots[0, 0]) == (0, 30)
    assert rew.tolist() == [np.float32(0.3), 0.0]


def test_pot_timer_runs_only_when_full(oracle_mod):
    env = make(oracle_mod, [(2, 2, D_UP, NONE), (3, 3, D_UP, NONE)])
    env.pots[0, 0] = (2, 30)
    for _ in range(5):
        step(env, [NOOP, NOOP])
    assert tuple(env.pots[0, 0]) == (2, 30)
    env.pots[0, 0] = (3, 30)
    for k in range(1, 31):
        step(env, [NOOP, NOOP])
        assert tuple(env.pots[0, 0]) == (3, 30 - k)
    step(env, [NOOP, NOOP])
    assert tuple(env.pots[0, 0]) == (3, 0)  # stays done until somebody plates it


def test_delivery_zone_takes_only_soup(oracle_mod):
    # the delivery zone '@' is at (4,4); an agent at (3,4) facing down is in front of it
    for held, kept, reward in ((ONION, ONION, 0.0), (PLATE, PLATE, 0.0), (SOUP, NONE, 1.0)):
        env = make(oracle_mod, [(3, 4, D_DOWN, held), (2, 2, D_UP, NONE)])
        _, rew, _, _ = step(env, [INTERACT, NOOP])
        assert env.agents[0, 0, 3] == kept
        assert rew.tolist() == [reward, reward]  # a delivery pays every agent


def test_closest_items_are_ordered_by_distance_then_cell(oracle_mod):
    """Four onions lie around; agent 0 at (2,3) must list them nearest first (L1), the two at
    equal distance in row-major cell order, each as (agent - item)."""
    env = make(oracle_mod, [(2, 3, D_UP, NONE), (3, 4, D_UP, NONE)])
    cells = {(1, 4): ONION, (3, 1): ONION, (2, 2): ONION, (1, 1): ONION, (3, 5): PLATE}
    for (r, c), it in cells.items():
        env.items[0, r * W + c] = it
    obs = env.featurize()[0]
    # distances from (2,3): (2,2) -> 1, (1,4) -> 2, (3,1) -> 3, (1,1) -> 3; tie: (1,1) is the earlier cell
    want = [(0, 1), (1, -1), (1, 2), (-1, 2)]
    assert obs[27:35].tolist() == [float(v) for pair in want for v in pair]
    assert obs[35:37].tolist() == [-1.0, -2.0] and not obs[37:43].any()  # one plate at (3,5), rest empty
    assert not obs[51:59].any()  # no soup anywhere
    # the agent can move left? no: an onion lies on (2,2); right (2,4) is free, down (3,3) free, up is the pot
    assert obs[97:101].tolist() == [1.0, 0.0, 1.0, 0.0]  # right, left, down, up
