"""Centralised-critic variant, interface of the reference's ``agents/CentralizedMAPPO.py``.

Differences from :class:`MAPPO`, all taken from the reference:
  * acting and bootstrapping evaluate ``centralised_critics`` on the concatenation of EVERY
    row's observation, width D*A*E (agents/MAPPO.py:70-73,127-131; model.py:29,45);
  * GAE uses, for every row, the mean reward over all rows of that step
    (agents/CentralizedMAPPO.py:77);
  * the value loss regresses onto the minibatch mean of the targets
    (agents/CentralizedMAPPO.py:36-46); as in the reference the value that gets trained
    there is the decentralised head's, because ``update`` evaluates the policy without
    ``joint_obs`` (agents/MAPPO.py:170).
With several ranks the per-step reward mean is formed over all ranks' rows.
"""
from __future__ import annotations

import torch

from .. import dist_utils
from ..gae import compute_gae as _gae_kernel
from ..gae import row_sums
from .mappo import MAPPO


class CMAPPO(MAPPO):
    centralised = True
    fused_target = "minibatch_mean"

    def compute_value_loss(self, target, new_values):
        centralized_target = target.mean(dim=0, keepdim=True)
        return 0.5 * ((new_values - centralized_target) ** 2).mean()

    def _bootstrap(self, next_obs):
        v = self.policy.get_value(next_obs, joint_obs=next_obs.view(1, -1))
        if v.shape != (1, 1):
            raise ValueError(f"centralised bootstrap has shape {tuple(v.shape)}, expected (1, 1)")
        return v

    def compute_gae(self, rewards, dones, values, next_values, want_returns=False):
        reward_mean = None
        if dist_utils.collectives_active():
            sums = dist_utils.all_reduce_sum_(row_sums(rewards))
            rows = torch.tensor([rewards.shape[1]], dtype=torch.float64, device=rewards.device)
            dist_utils.all_reduce_sum_(rows)
            reward_mean = (sums / rows).float()
        adv, ret = _gae_kernel(rewards, dones, values, next_values, self.gamma, self.lam, centralised=True,
                               reward_mean=reward_mean, want_returns=want_returns)
        return (adv, ret) if want_returns else adv


CMAPPO._value_loss_of_fused_pass = CMAPPO.compute_value_loss
