"""How the centralised GAE variant is held to the reference's vectors (shared by the CPU and GPU suites)."""
import numpy as np

RTOL = 1e-5  # the tolerance the task statement allows for fp32 GAE


def exact_centralised(g, k):
    """CentralizedMAPPO.compute_gae (agents/CentralizedMAPPO.py:48-82) in float64 on the float32 inputs:
    what both float32 programs approximate."""
    gamma, lam = (float(x) for x in g[k + "hp"])
    r, d, v = (g[k + n].astype(np.float64) for n in ("rewards", "dones", "values"))
    nv = np.asarray(g[k + "next_values"], np.float64).reshape(-1)
    T = r.shape[0]
    adv, last = np.zeros_like(r), 0.0
    for t in range(T - 1, -1, -1):
        mask, nxt = (1 - d[t], nv) if t == T - 1 else (1 - d[t + 1], v[t + 1])
        last = r[t].mean() + gamma * nxt * mask - v[t] + gamma * lam * mask * last
        adv[t] = last
    return adv, adv + v


def check_centralised(got, want, exact, what):
    """The centralised variant differs from the reference's bits only through the per-step reward mean
    (summed in double and rounded once here, in torch's float32 order there).  Three statements:
      1. every entry that is not a near-cancellation (|x| >= 2 % of the tensor's largest) agrees to
         1e-5 relative, no absolute term;
      2. the rest (results of subtracting nearly equal O(1) operands) agrees to 4 float32 ulp of the
         tensor's scale;
      3. measured against float64 arithmetic on the same inputs, this implementation is no farther
         away than the reference's own float32 output is.
    Returns the measured figures."""
    scale = np.abs(want).max()
    big = np.abs(want) >= 2e-2 * scale
    rel = float((np.abs(got - want)[big] / np.abs(want)[big]).max()) if big.any() else 0.0
    small = float(np.abs(got - want)[~big].max() / scale) if (~big).any() else 0.0
    assert rel <= RTOL, f"{what}: max relative error {rel:.3e} over entries >= 2 % of the scale"
    assert small <= 4 * 2.0 ** -23, f"{what}: near-zero entries off by {small:.3e} of the scale"
    ours, theirs = np.abs(got - exact).max(), np.abs(want - exact).max()
    assert ours <= theirs * 1.05 + 2.0 ** -24 * scale, f"{what}: farther from float64 ({ours:.3e}) than the reference is ({theirs:.3e})"
    wide = np.abs(want) > 1e-3 * scale
    rel_wide = float((np.abs(got - want)[wide] / np.abs(want)[wide]).max())
    ref_wide = float((np.abs(want - exact)[wide] / np.abs(exact)[wide]).max())
    return {"rel_2pct": rel, "near_zero_over_scale": small, "rel_0.1pct": rel_wide, "reference_vs_float64_0.1pct": ref_wide}
