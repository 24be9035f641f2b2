"""A stand-in with the attribute surface of a cogrid Overcooked env, for exercising
tools/record_cogrid_episode.py where cogrid is absent: ``env.agent_ids``, ``env.reset()``,
``env.step({id: action})`` with per-agent dicts, ``env.grid.grid_agents[id].pos/.dir/.inventory``,
``env.grid.get(r, c)``, ``env.grid.grid`` (objects with ``.pos`` / ``.obj_placed_on``), pots with
``objects_in_pot`` / ``cooking_timer``.  The dynamics behind it are this repository's CPU oracle
(one env), so a recording made from it must replay exactly."""
import numpy as np

OBS_KEY = "n_agent_overcooked_features"


def _cls(name, **attrs):
    return type(name, (), attrs)


Wall, OnionStack, PlateStack, DeliveryZone = (_cls(n) for n in ("Wall", "OnionStack", "PlateStack", "DeliveryZone"))
Onion, Plate, OnionSoup = (_cls(n) for n in ("Onion", "Plate", "OnionSoup"))
Counter, Pot, GridAgent = _cls("Counter", obj_placed_on=None), _cls("Pot"), _cls("GridAgent")
ITEMS = {1: Onion, 2: Plate, 3: OnionSoup}
STATIC = {"#": Wall, "C": Counter, "U": Pot, "O": OnionStack, "=": PlateStack, "@": DeliveryZone}


class StubGrid:
    def __init__(self, rows, ora):
        self.rows, self.ora = rows, ora
        self.height, self.width = len(rows), len(rows[0])
        self.refresh()

    def refresh(self):
        H, W, o = self.height, self.width, self.ora
        self.grid, pot = [], 0
        for r in range(H):
            for c in range(W):
                ch, lying = self.rows[r][c], int(o.items[0, r * W + c])
                obj = STATIC[ch]() if ch in STATIC else (ITEMS[lying]() if lying else None)
                if ch == "C" and lying:
                    obj.obj_placed_on = ITEMS[lying]()
                if ch == "U":
                    obj.objects_in_pot = [Onion() for _ in range(int(o.pots[0, pot, 0]))]
                    obj.cooking_timer = int(o.pots[0, pot, 1])
                    pot += 1
                if obj is not None:
                    obj.pos = (r, c)
                self.grid.append(obj)
        self.grid_agents = {}
        for a in range(o.A):
            ag = GridAgent()
            r, c, d, inv = (int(x) for x in o.agents[0, a])
            ag.pos, ag.dir, ag.inventory = (r, c), d, ([ITEMS[inv]()] if inv else [])
            self.grid_agents[a] = ag

    def get(self, r, c):
        return self.grid[r * self.width + c]


class StubOvercooked:
    def __init__(self, oracle_mod, rows, num_agents, feature, max_steps, seed):
        self.ora = oracle_mod.OracleVecEnv(rows, num_agents, feature, max_steps=max_steps, num_envs=1, seed=seed,
                                           auto_reset=False)
        self.rows, self.agent_ids = list(rows), list(range(num_agents))
        self.grid = None

    def _dicts(self, obs):
        return {i: {OBS_KEY: obs[i].copy()} for i in self.agent_ids}

    def reset(self):
        obs = self.ora.reset()
        self.grid = StubGrid(self.rows, self.ora)
        return self._dicts(obs), {i: {} for i in self.agent_ids}

    def step(self, actions):
        act = np.array([actions[i] for i in self.agent_ids], np.int64)
        obs, rew, term, trunc = self.ora.step(act)
        self.grid.refresh()
        ids = self.agent_ids
        return (self._dicts(obs), {i: float(rew[i]) for i in ids}, {i: bool(term[i]) for i in ids},
                {i: bool(trunc[i]) for i in ids}, {i: {} for i in ids})
