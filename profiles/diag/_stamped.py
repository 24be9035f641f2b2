"""For the -DOC_STAMPS diagnostic build only: there the priority-override argument of marl_env_step
carries the stamp buffer (uint64 [waves][16]), which the product's step_into would refuse by its size."""
from marl_amd import _lib
from marl_amd.env import _ptr


def stamped_step(env, actions, obs, rew, term, trunc, dbg):
    _lib.check(env._L.marl_env_step(env._h, _ptr(env.state), _ptr(actions), actions.element_size(), _ptr(obs), _ptr(rew),
                                    _ptr(term), _ptr(trunc), _ptr(dbg), env._stream()), "marl_env_step")
