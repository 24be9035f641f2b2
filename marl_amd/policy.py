"""Actor-critic MLP with the parameter names of the reference's ``model.Agent``
(model.py:11-68), so that checkpoints interchange: ``network.{0,2}`` (obs -> 512 -> 256,
ReLU), ``actor`` (256 -> n_actions), ``critic`` (256 -> 1) and ``centralised_critics.{0,2,4}``
(obs_dim * num_agents * num_envs -> 512 -> 256 -> 1).  All layers start orthogonal with gain
sqrt(2) and zero bias (model.py:6-9).  The matrix products run on PyTorch-ROCm (rocBLAS /
hipBLASLt); nothing here is hand-written HIP; which library call computes a layer of a small
batch is a fixed function of its shape (`_table_form`)."""
from __future__ import annotations

import math
import os
import threading
import warnings

import torch
from torch import nn
from torch.distributions import Categorical


def _ortho(linear: nn.Linear, gain: float = math.sqrt(2.0), bias: float = 0.0) -> nn.Linear:
    nn.init.orthogonal_(linear.weight, gain)
    nn.init.constant_(linear.bias, bias)
    return linear


def _weight_grad(gy, x):
    """gy^T @ x for a tall batch (K rows, K >> features).  hipBLASLt's kernel choice for this
    skinny shape leaves most of the chip idle (1.15 ms for 512 x 202 over K = 262,144); cutting K
    into S slabs, one batched GEMM and a sum over the slabs is 2-4x faster on MI355X
    (0.50 ms for the same shape).  Same float32 products, another summation order."""
    K = gy.shape[0]
    S = 1
    while S < 128 and K % (2 * S) == 0 and K // (2 * S) >= 2048:
        S *= 2
    if S == 1 or not gy.is_cuda:
        return gy.t() @ x
    return torch.bmm(gy.view(S, K // S, -1).transpose(1, 2), x.view(S, K // S, -1)).sum(0)


# Below 512 rows (the reference's batch sizes: 16 envs x 2 agents = 32 rows per rollout step)
# hipBLASLt's pick for the fused call is erratic on MI355X: rows x 512 -> 256 takes 17.8 us at 32
# rows, 60 us at 128, 118 us at 256 and 6 us at 512, while a plain rocBLAS product takes 3.7 us at
# all of them (plus two small in-place launches for bias and ReLU); for the other layers of the same
# MLP the fused call wins (rows x 202 -> 512: 5.4 against 7.5 us).  (profiles/diag/small_gemm2.py)
# The forms sum in different orders, so which one runs must not depend on anything that varies from
# run to run or from rank to rank: it is a fixed function of the layer's shape (`_TABLE_FORM`).
# MARL_AMD_SMALL_GEMM=fused|plain|gemv forces one form; MARL_AMD_SMALL_GEMM=tune times the forms once
# per shape instead (a tuning aid: its choices can differ between runs, and with them the bits of a
# seeded run).
_SMALL_ROWS = 512
_GEMM_CHOICE = {}
_BLAS_LOCK = threading.Lock()


def _table_form(rows, in_features, out_features):
    """the measured winners on MI355X: a single row (the centralised critic's joint observation) as a
    matrix-vector product; the 512-wide hidden layer through rocBLAS; everything else fused"""
    if rows == 1 and in_features >= 1024:
        return "gemv"
    if in_features == 512 and out_features <= 256:
        return "plain"
    return "fused"


def gemm_choices():
    """the forms in use so far, {(rows, in_features, out_features, relu): "fused" | "plain" | "gemv"}"""
    return dict(_GEMM_CHOICE)


def _fused_linear(x, weight, bias, relu):
    if relu:
        return torch._addmm_activation(bias, x, weight.t(), use_gelu=False)
    return torch.addmm(bias, x, weight.t())


_blas_switch_announced = False


class _FirstBlasSwitch(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        blas = torch.backends.cuda.preferred_blas_library
        blas(blas())
        return t


def _announce_blas_switch_quietly(t):
    """PyTorch issues a one-time "experimental feature" notice when the preferred BLAS library is
    set: as a Python warning when the call happens inside an operator (raised only when the
    outermost operator returns, here `_LinearReLU.apply`, where it can no longer be filtered at its
    source), straight to stderr otherwise.  The first switch is therefore made inside a small
    operator of its own with warnings ignored around it."""
    global _blas_switch_announced
    if not _blas_switch_announced:
        _blas_switch_announced = True
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            _FirstBlasSwitch.apply(t)


def _plain_linear(x, weight, bias, relu):
    """the product through rocBLAS, bias and ReLU as in-place launches of their own"""
    blas = torch.backends.cuda.preferred_blas_library
    with _BLAS_LOCK:  # the preference is process-wide: one thread at a time, always put back
        prev = blas()
        blas("hipblas")
        try:
            out = torch.mm(x, weight.t())
        finally:
            blas(prev)
    out.add_(bias)
    return out.relu_() if relu else out


def _gemv_linear(x, weight, bias, relu):
    """a single row (the centralised critic's joint observation, model.py:45): matrix-vector product"""
    out = torch.addmv(bias, weight, x[0]).unsqueeze(0)
    return out.relu_() if relu else out


_FORMS = {"fused": _fused_linear, "plain": _plain_linear, "gemv": _gemv_linear}


def _small_gemm_form(x, weight, bias, relu):
    rows = x.shape[0]
    if rows >= _SMALL_ROWS:
        return "fused"
    key = (rows, weight.shape[1], weight.shape[0], bool(relu))
    form = _GEMM_CHOICE.get(key)
    if form is not None:
        return form
    forced = os.environ.get("MARL_AMD_SMALL_GEMM", "table")
    if forced in ("fused", "plain", "gemv"):
        form = forced if (forced != "gemv" or rows == 1) else "fused"
    elif forced != "tune":
        form = _table_form(rows, weight.shape[1], weight.shape[0])
    elif torch.cuda.is_current_stream_capturing():
        return "fused"  # cannot time under capture; decided at the next eager call
    else:
        # Timed as the small-batch rollouts and updates run them: replayed from a hipGraph.  (Launched
        # eagerly from Python both forms are bound by the host's launch rate and the form with fewer
        # launches always looks better, whatever its kernels cost.)
        times = []
        names = ("fused", "plain", "gemv") if rows == 1 else ("fused", "plain")
        for name in names:
            fn = _FORMS[name]
            fn(x, weight, bias, relu)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                for _ in range(10):
                    fn(x, weight, bias, relu)
            g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                g.replay()
            e1.record()
            e1.synchronize()
            times.append(e0.elapsed_time(e1))
            del g
        # another form has to win clearly: a close call must not flip from run to run
        best = min(range(len(times)), key=lambda k: times[k])
        form = names[best] if times[best] * 1.25 < times[0] else "fused"
    _GEMM_CHOICE[key] = form
    return form


class _LinearReLU(torch.autograd.Function):
    """relu(x @ W^T + b) (or without the ReLU): one GEMM with a fused epilogue, or for the small
    batches where that call is slow a plain product plus in-place bias and ReLU; the backward
    pass forms the weight gradient with `_weight_grad`."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        out = _FORMS[_small_gemm_form(x, weight, bias, relu)](x, weight, bias, relu)
        ctx.relu = relu
        ctx.save_for_backward(x, weight, out if relu else None)
        return out

    @staticmethod
    def backward(ctx, gy):
        x, weight, out = ctx.saved_tensors
        if ctx.relu:
            gy = torch.ops.aten.threshold_backward(gy, out, 0)
        gy = gy.contiguous()
        gx = gy @ weight if ctx.needs_input_grad[0] else None
        return gx, _weight_grad(gy, x), gy.sum(0), None


def _linear(layer: nn.Linear, x, relu):
    if x.is_cuda and x.dim() == 2:
        if x.shape[0] < _SMALL_ROWS:
            _announce_blas_switch_quietly(x)
        return _LinearReLU.apply(x.contiguous(), layer.weight, layer.bias, relu)
    out = layer(x)
    return torch.relu(out) if relu else out


def _mlp(sizes, final_activation):
    layers = []
    for k in range(len(sizes) - 1):
        layers.append(_ortho(nn.Linear(sizes[k], sizes[k + 1])))
        if k < len(sizes) - 2 or final_activation:
            layers.append(nn.ReLU())
    return nn.Sequential(*layers)


class Agent(nn.Module):
    def __init__(self, obs_space, action_space, num_agents=4, num_envs=1, build_centralised=True):
        super().__init__()
        self.obs_space, self.action_space = obs_space, action_space
        self.num_agents, self.num_envs = int(num_agents), int(num_envs)
        d = int(obs_space.shape[0])
        self.network = _mlp([d, 512, 256], final_activation=True)
        self.actor = _ortho(nn.Linear(256, int(action_space.n)))
        self.critic = _ortho(nn.Linear(256, 1))
        # model.py:28-34: one value for the concatenated observations of every row
        # Its first layer is obs_dim * num_agents * num_envs wide: 54 GB of weights at 65,536 envs, and
        # an orthogonal initialisation (a QR factorisation) of that matrix on the host.  A
        # decentralised learner never evaluates it and may leave it out.
        self.sampler = None  # set by use_fused_sampling(); plain torch.distributions otherwise
        self._head_w = self._head_b = None  # stacked heads of rollout_step
        self.centralised_critics = (_mlp([d * self.num_agents * self.num_envs, 512, 256, 1], final_activation=False)
                                    if build_centralised else None)

    def _trunk(self, x):
        """network(x): each Linear + ReLU pair runs as one hipBLASLt GEMM with the bias and the
        ReLU in its epilogue: the same float32 result bit for bit, one pass less over the
        activations (250 us instead of 365 us for 131,072 x 202 -> 512)."""
        return _linear(self.network[2], _linear(self.network[0], x, True), True)

    def use_fused_sampling(self, seed, row_index_base=0):
        """Rollout actions (no autograd, action=None) come from `sampling.CategoricalSampler`: one
        launch for sample + log_prob + entropy, its own counter RNG instead of torch's generator."""
        from .sampling import CategoricalSampler

        self.sampler = CategoricalSampler(seed, row_index_base, next(self.parameters()).device)
        return self

    def _central(self, joint_obs):
        if self.centralised_critics is None:
            raise RuntimeError("this Agent was built without its centralised critic (build_centralised=False)")
        c = self.centralised_critics
        return _linear(c[4], _linear(c[2], _linear(c[0], joint_obs, True), True), False)

    def get_value(self, x, joint_obs=None):
        if joint_obs is not None:
            want = int(self.obs_space.shape[0]) * self.num_agents * self.num_envs
            if joint_obs.shape[1] != want:
                raise ValueError(f"joint_obs has width {joint_obs.shape[1]}, expected {want}")
            return self._central(joint_obs)
        return _linear(self.critic, self._trunk(x), False)

    def get_action_and_value(self, x, action=None, joint_obs=None):
        hidden = self._trunk(x)
        # validate_args=False: the argument check reads a device flag back on the host, one
        # synchronisation per call, and cannot run under hipGraph capture; it changes no result
        logits = _linear(self.actor, hidden, False)
        if action is None and self.sampler is not None and logits.is_cuda and not torch.is_grad_enabled():
            # rollout: sample, log-probability and entropy in one launch
            value = self._central(joint_obs) if joint_obs is not None else _linear(self.critic, hidden, False)
            action, logprob, entropy = self.sampler.sample(logits)
            return action, logprob, entropy, value
        dist = Categorical(logits=logits, validate_args=False)
        if action is None:
            action = dist.sample()
        value = self._central(joint_obs) if joint_obs is not None else _linear(self.critic, hidden, False)
        return action, dist.log_prob(action), dist.entropy(), value

    def refresh_rollout_heads(self):
        """Actor and critic heads stacked into one [actions + 1, hidden] weight for `rollout_step`
        (same storage every time, so captured launches keep reading the right place).  Call after
        the parameters changed."""
        k = self.actor.out_features
        if getattr(self, "_head_w", None) is None or self._head_w.device != self.actor.weight.device:
            self._head_w = torch.empty((k + 1, self.actor.in_features), device=self.actor.weight.device)
            self._head_b = torch.empty(k + 1, device=self.actor.weight.device)
        with torch.no_grad():
            torch.cat([self.actor.weight, self.critic.weight], out=self._head_w)
            torch.cat([self.actor.bias, self.critic.bias], out=self._head_b)

    def rollout_step(self, x, logprob_out, value_out, actions_out=None, slot=None, joint=False, row_offset=0):
        """The rollout half of `get_action_and_value` for the decentralised critic, writing the
        log-probabilities and the values straight into the given rows of the rollout store
        (float32 [rows] each) instead of returning fresh tensors that are then copied.  Both
        heads come out of one GEMM (`refresh_rollout_heads`), the sampling kernel hands the value
        column on: three launches after the trunk instead of five.  The numbers agree with
        `get_action_and_value` to float32 rounding; needs `use_fused_sampling`, no autograd.
        `joint=True`: the value is the centralised critic's, one number for the concatenation of all
        rows (agents/MAPPO.py:70-73, model.py:45), written to every row as the reference's buffer
        assignment broadcasts it (agent_environment.py:83-84, buffer.py:28).
        `row_offset`: `x` holds rows [row_offset, row_offset + len(x)) of the batch (one shard of a
        pipelined rollout); they draw the random numbers of those rows.  Not with `joint`."""
        if joint and row_offset:
            raise ValueError("the centralised critic reads every row of the batch: no per-shard call")
        if self.sampler is None:
            raise RuntimeError("rollout_step needs use_fused_sampling()")
        if joint:
            with torch.no_grad():
                logits = _linear(self.actor, self._trunk(x), False)
                actions, _, _ = self.sampler.sample(logits, want_entropy=False, actions_out=actions_out,
                                                    logprob_out=logprob_out, slot=slot)
                value_out.copy_(self._central(x.view(1, -1)).view(1).expand_as(value_out))
            return actions
        if getattr(self, "_head_w", None) is None:
            self.refresh_rollout_heads()
        with torch.no_grad():
            heads = torch.addmm(self._head_b, self._trunk(x), self._head_w.t())
            actions, _, _ = self.sampler.sample(heads, want_entropy=False, actions_out=actions_out, logprob_out=logprob_out,
                                                slot=slot, num_actions=self.actor.out_features, next_column_out=value_out,
                                                row_offset=row_offset)
        return actions

    def logits_and_value(self, x):
        """both heads on the shared trunk (model.py:49-61 without the distribution object)"""
        hidden = self._trunk(x)
        return _linear(self.actor, hidden, False), _linear(self.critic, hidden, False)

    def get_prob(self, x):
        return Categorical(logits=_linear(self.actor, self._trunk(x), False), validate_args=False)
