"""Multi-agent PPO learner with decentralised critic, interface of the reference's
``agents/MAPPO.py`` (act / add_to_buffer / compute_gae / update).

What changed relative to the reference, and only this:
  * ``compute_gae`` runs as one HIP kernel (libmarl_amd.so) instead of a Python loop over T
    (agents/MAPPO.py:98-114); value targets (``:144``) come out of the same launch;
  * minibatch indices are drawn on the device (``torch.randperm``) unless
    ``shuffle="numpy"`` asks for the reference's ``np.random.shuffle`` (``:159``);
  * with more than one rank, gradients are averaged by one flat all-reduce between
    ``loss.backward()`` and ``clip_grad_norm_`` (``:207-208``);
  * scalar logging does not force a device sync per minibatch.
The PPO objective itself (``:170-209``: clipped surrogate, unclipped value loss, entropy bonus,
no advantage normalisation) is the reference's.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from .. import dist_utils
from ..gae import compute_gae as _gae_kernel


class MAPPO:
    centralised = False
    fused_target = "own"  # what the one-pass loss regresses a sample's value onto: its own target

    def __init__(self, env, optimzer, policy, buffer, single_agent_obs, single_agent_action, batch_size=1,
                 num_mini_batches=4, num_agents=4, ppo_epoch=10, clip_param=0.2, value_loss_coef=0.5,
                 entropy_coef=0.01, max_grad_norm=0.5, gamma=0.99, lam=0.95, save_path=None, log_dir=None,
                 log=False, args=None, shuffle="device", writer=None, graph_update="off", fused_loss=True):
        self.env, self.optimizer, self.policy, self.buffer = env, optimzer, policy, buffer
        self.num_agents = num_agents
        self.single_agent_obs, self.single_agent_action = single_agent_obs, single_agent_action
        self.device = next(policy.parameters()).device if policy is not None else torch.device(
            "cuda" if torch.cuda.is_available() else "cpu")
        self.batch_size = batch_size
        self.mini_batch_size = batch_size // num_mini_batches
        self.ppo_epoch, self.clip_param = ppo_epoch, clip_param
        self.value_loss_coef, self.entropy_coef, self.max_grad_norm = value_loss_coef, entropy_coef, max_grad_norm
        self.gamma, self.lam = gamma, lam
        self.save_path, self.log_dir, self.args, self.log = save_path, log_dir, args, log
        self.shuffle = shuffle
        self.summary_writer = writer
        self.num_gradient_steps = 0
        self.last_stats = {}
        self._reducer = dist_utils.FlatGradAllReducer(policy.parameters()) if policy is not None else None
        # "auto": replay the minibatch step (forward, backward, clip, Adam) from a hipGraph when it is
        # launch-bound; single rank only (the gradient all-reduce sits in the middle of the step)
        small = batch_size // max(num_mini_batches, 1) <= 65536
        self._graph_update = (graph_update == "on" or (graph_update == "auto" and small)) \
            and not dist_utils.collectives_active() and self.device.type == "cuda" and optimzer is not None
        self._g = None
        # loss + head gradients from one HIP pass (marl_amd.ppo_loss) instead of ~70 autograd ops, on
        # the GPU, for the value losses the pass knows: this class's and CMAPPO's (which regresses
        # onto the minibatch mean of the targets: the same pass with that mean as every sample's
        # target).  A subclass with another compute_value_loss falls back to autograd.
        self._fused_loss = bool(fused_loss) and self.device.type == "cuda" \
            and type(self).compute_value_loss is getattr(type(self), "_value_loss_of_fused_pass", None)
        self._mean_target = None
        if self._graph_update:
            for group in optimzer.param_groups:  # the step counter must live on the device to be captured
                group["capturable"] = True

    # ------------------------------------------------------------------ acting
    def act(self, obs):
        """agents/MAPPO.py:58-77: sample actions without tracking gradients."""
        with torch.no_grad():
            obs = obs.to(self.device)
            if self.centralised:
                return self.policy.get_action_and_value(obs, joint_obs=obs.view(1, -1))
            return self.policy.get_action_and_value(obs)

    def act_into(self, obs, step, actions_out, rows=None):
        """`act` + the learner-side half of `add_to_buffer` for row `step` of the rollout store:
        log-probabilities and values are written into their rows by the kernels that produce them,
        the int64 actions into `actions_out` (the env reads them there; the driver converts a
        whole rollout of them to the store's float32 at once, `store_actions`).  Call `step` of a
        rollout draws with call number rollout * num_steps + step.  Needs fused sampling; with the
        centralised critic every row receives the one joint value.  `rows` (a slice): `obs` and
        `actions_out` are that slice of the batch's rows (one shard of a pipelined rollout,
        decentralised critic only); results land in the same slice of the store's rows."""
        buf = self.buffer
        if rows is None:
            return self.policy.rollout_step(obs, buf.logprobs_buff[step], buf.values_buff[step], actions_out, slot=step,
                                            joint=self.centralised)
        return self.policy.rollout_step(obs, buf.logprobs_buff[step][rows], buf.values_buff[step][rows], actions_out,
                                        slot=step, joint=self.centralised, row_offset=rows.start)

    def can_act_on_shards(self):
        """per-shard `act_into` calls: fused sampling and a critic that reads only a row's own observation"""
        return self.can_act_into() and not self.centralised

    def begin_direct_rollouts(self, num_steps):
        self.policy.sampler.use_slots(num_steps)

    def begin_direct_rollout(self):
        self.policy.refresh_rollout_heads()  # the update changed the parameters

    def end_direct_rollout(self, actions_rows):
        """after the last step of a rollout: the actions as the reference stores them (float32,
        buffer.py:12) and the call numbers of the next rollout"""
        self.buffer.actions_buff.copy_(actions_rows)
        self.policy.sampler.advance_round()

    def can_act_into(self):
        return getattr(self.policy, "sampler", None) is not None

    def add_to_buffer(self, obs, actions, rewards, dones, logprobs, values):
        self.buffer.add(obs, actions, rewards, dones, logprobs, values)

    # ------------------------------------------------------------------ GAE
    def compute_gae(self, rewards, dones, values, next_values, want_returns=False):
        """agents/MAPPO.py:82-114.  Returns advantages (and value targets when asked)."""
        adv, ret = _gae_kernel(rewards, dones, values, next_values, self.gamma, self.lam, centralised=False,
                               want_returns=want_returns)
        return (adv, ret) if want_returns else adv

    def compute_value_loss(self, target, newvalue):
        return 0.5 * ((newvalue - target) ** 2).mean()  # agents/MAPPO.py:243-248

    def _bootstrap(self, next_obs):
        rows = self.buffer.rewards_buff.shape[1]
        return self.policy.get_value(next_obs).reshape(1, rows)  # agents/MAPPO.py:133

    # ------------------------------------------------------------------ update
    def _minibatch_step(self, mb, b_obs, b_actions_f, b_logprobs, b_advantages, b_value_target):
        """agents/MAPPO.py:167-209 for one minibatch; returns the seven logged statistics."""
        if self._fused_loss:
            from ..ppo_loss import ppo_loss

            logits, newvalue = self.policy.logits_and_value(b_obs[mb])
            if self.fused_target == "minibatch_mean":
                # agents/CentralizedMAPPO.py:36-46: every sample regresses onto the mean target of its minibatch
                # (the mean of THIS rank's minibatch, as in the reference's one process; the pass reads
                # the target of a sample at the sample's index, so only those entries are written)
                if self._mean_target is None or self._mean_target.shape != b_value_target.shape:
                    self._mean_target = torch.empty_like(b_value_target)
                self._mean_target.index_copy_(0, mb, b_value_target[mb].mean().expand(mb.numel()))
                b_value_target = self._mean_target
            loss, stats = ppo_loss(logits, newvalue, mb, b_actions_f, b_logprobs, b_advantages, b_value_target,
                                   self.clip_param, self.entropy_coef, self.value_loss_coef)
            self.optimizer.zero_grad()
            loss.backward()
            self._reducer()
            norm = torch.nn.utils.clip_grad_norm_(self.policy.parameters(), self.max_grad_norm)
            self.optimizer.step()
            return torch.cat([stats[1:], self._clipped_norm(norm).reshape(1)])
        _, newlogprob, entropy, newvalue = self.policy.get_action_and_value(b_obs[mb], b_actions_f[mb].long())
        logratio = newlogprob - b_logprobs[mb]
        ratio = logratio.exp()
        mb_adv = b_advantages[mb]
        pg_loss = torch.max(-mb_adv * ratio,
                            -mb_adv * torch.clamp(ratio, 1.0 - self.clip_param, 1.0 + self.clip_param)).mean()
        v_loss = self.compute_value_loss(b_value_target[mb], newvalue.squeeze())
        entropy_loss = entropy.mean()
        loss = pg_loss - self.entropy_coef * entropy_loss + v_loss * self.value_loss_coef

        self.optimizer.zero_grad()
        loss.backward()
        self._reducer()  # RCCL: one flat all-reduce, no-op on a single rank
        norm = torch.nn.utils.clip_grad_norm_(self.policy.parameters(), self.max_grad_norm)
        self.optimizer.step()

        with torch.no_grad():
            return torch.stack([v_loss, pg_loss, entropy_loss, ((ratio - 1) - logratio).mean(), (-logratio).mean(),
                                ((ratio - 1.0).abs() > self.clip_param).float().mean(),
                                self._clipped_norm(norm).float()]).double()

    def _clipped_norm(self, total_norm):
        """The reference logs the gradient norm AFTER clip_grad_norm_ has scaled the gradients in
        place (agents/MAPPO.py:208,216): the norm before clipping times the factor the clip applied
        (max_norm / (norm + 1e-6), at most 1).  Computed from clip_grad_norm_'s return value, on the
        device, without another pass over the gradients."""
        with torch.no_grad():
            scale = torch.clamp(self.max_grad_norm / (total_norm + 1e-6), max=1.0)
            return (total_norm * scale).double()

    def _replay_minibatch(self, mb, b_obs, b_actions_f, b_logprobs, b_advantages, b_value_target):
        """The same step replayed from a hipGraph.  Captured once, after the first eager update has
        created the optimizer state; inputs are read from fixed addresses (the rollout store and
        three staging tensors), so a replay needs one index copy and one launch."""
        g = self._g
        if g is None or g["obs"] != b_obs.data_ptr():
            total = b_logprobs.numel()
            st = {"obs": b_obs.data_ptr(),
                  "mb": torch.zeros(self.mini_batch_size, dtype=torch.int64, device=self.device),
                  "adv": torch.empty(total, dtype=torch.float32, device=self.device),
                  "ret": torch.empty(total, dtype=torch.float32, device=self.device)}
            st["adv"].copy_(b_advantages); st["ret"].copy_(b_value_target)
            st["update"] = self.num_gradient_steps
            st["mb"].copy_(mb)
            torch.cuda.current_stream(self.device).synchronize()
            st["graph"] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(st["graph"], capture_error_mode="thread_local"):
                st["out"] = self._minibatch_step(st["mb"], b_obs, b_actions_f, b_logprobs, st["adv"], st["ret"])
            self._g = g = st
        if g["update"] != self.num_gradient_steps:  # a new update: advantages and targets of this rollout
            # (never keyed on the tensors' addresses: the allocator hands the same block out again)
            g["adv"].copy_(b_advantages); g["ret"].copy_(b_value_target)
            g["update"] = self.num_gradient_steps
        g["mb"].copy_(mb)
        g["graph"].replay()
        return g["out"]

    def update(self, next_obs):
        """agents/MAPPO.py:116-233."""
        buf = self.buffer
        with torch.no_grad():
            next_values = self._bootstrap(next_obs.to(self.device)).float()
        advantages, value_target = self.compute_gae(buf.rewards_buff, buf.dones_buff, buf.values_buff, next_values,
                                                    want_returns=True)
        obs_shape = tuple(self.env.single_observation_space.shape)
        b_obs = buf.obs_buff.reshape((-1,) + obs_shape)
        b_logprobs = buf.logprobs_buff.reshape(-1)
        b_actions_f = buf.actions_buff.reshape(-1)  # stored as float32 like the reference's buffer
        b_advantages = advantages.reshape(-1)
        b_value_target = value_target.reshape(-1)
        total = b_logprobs.numel()
        if self.batch_size != total:
            raise ValueError(f"batch_size {self.batch_size} != stored samples {total}")

        sums = torch.zeros(7, dtype=torch.float64, device=self.device)  # v, pg, ent, kl, old_kl, clipfrac, grad norm
        count = 0
        idx = np.arange(total)  # agents/MAPPO.py:156: shuffled in place epoch after epoch, never re-sorted
        for _ in range(self.ppo_epoch):
            if self.shuffle == "numpy":
                np.random.shuffle(idx)
                perm = torch.from_numpy(idx).to(self.device)
            else:
                perm = torch.randperm(total, device=self.device)
            for start in range(0, total, self.mini_batch_size):
                mb = perm[start:min(start + self.mini_batch_size, total)]
                if self._graph_update and self.num_gradient_steps > 0 and mb.numel() == self.mini_batch_size:
                    sums += self._replay_minibatch(mb, b_obs, b_actions_f, b_logprobs, b_advantages, b_value_target)
                else:
                    sums += self._minibatch_step(mb, b_obs, b_actions_f, b_logprobs, b_advantages, b_value_target)
                count += 1
        self.num_gradient_steps += 1
        stats = (sums / max(count, 1)).tolist()  # one sync per update
        self.last_stats = dict(zip(("value_loss", "policy_loss", "entropy", "approx_kl", "old_approx_kl", "clipfrac",
                                    "policy_grad_norm"), stats))
        if self.log and self.summary_writer is not None:
            self.summary_writer.add_scalar("charts/learning_rate", self.optimizer.param_groups[0]["lr"],
                                           self.num_gradient_steps)
            for k, v in self.last_stats.items():
                tag = "NN/policy_grad_norm" if k == "policy_grad_norm" else f"losses/{k}"  # agents/MAPPO.py:213-218
                self.summary_writer.add_scalar(tag, v, self.num_gradient_steps)
        if self.save_path is not None and self.num_gradient_steps % 100 == 0 and dist_utils.get_rank() == 0:
            self.save_model()
        buf.reset()

    def checkpoint_name(self):
        a = self.args
        kind = "centralised" if getattr(a, "centralised", False) else "decentralised"  # agents/MAPPO.py:227-228
        return f"{kind}_policy_{a.num_agents}_agents_{a.layout}_seed_{a.seed}.pth"

    def save_model(self):
        """agents/MAPPO.py:226-231.  The reference's loader is strict (tests/test_load.py:71) and its
        model always holds a centralised critic obs_dim * num_agents * num_envs wide (model.py:28-34).
        A run with many envs does not build that critic here (gigabytes of never-trained weights);
        its checkpoint then lacks the centralised_critics.* entries and loads into the reference's
        model only with strict=False, which is said once when it is written."""
        os.makedirs(self.save_path, exist_ok=True)
        path = os.path.join(self.save_path, self.checkpoint_name())
        sd = self.policy.state_dict()
        if not any(k.startswith("centralised_critics.") for k in sd) and not getattr(self, "_warned_critic", False):
            self._warned_critic = True
            print(f"note: {path} holds no centralised_critics.* entries (the critic is not built at this env count); "
                  "the reference's model loads it with load_state_dict(..., strict=False)", flush=True)
        torch.save(sd, path)
        return path

    def get_grad_norm(self, module):
        sq = [p.grad.detach().norm(2) ** 2 for p in module.parameters() if p.grad is not None]
        return float(torch.stack(sq).sum().sqrt()) if sq else 0.0


MAPPO._value_loss_of_fused_pass = MAPPO.compute_value_loss
