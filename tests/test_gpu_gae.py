"""GAE kernel vs the reference-generated golden vectors and the CPU oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden", "gae_golden.npz")
RTOL = 1e-5  # the tolerance the task statement allows for fp32 GAE


from gae_checks import check_centralised, exact_centralised  # noqa: E402


def _run(g, k, mode):
    from marl_amd.gae import compute_gae

    gamma, lam = (float(x) for x in g[k + "hp"])
    dev = "cuda:0"
    adv, ret = compute_gae(torch.from_numpy(g[k + "rewards"]).to(dev), torch.from_numpy(g[k + "dones"]).to(dev),
                           torch.from_numpy(g[k + "values"]).to(dev),
                           torch.from_numpy(g[k + "next_values"]).to(dev).reshape(1, -1), gamma, lam,
                           centralised=bool(mode))
    return adv.cpu().numpy(), ret.cpu().numpy()


def test_gae_matches_reference_golden_vectors():
    g = np.load(GOLD)
    for ci in range(int(g["num_cases"])):
        for mode in (0, 1):
            k = f"c{ci}_m{mode}_"
            adv, ret = _run(g, k, mode)
            if mode == 0:
                # decentralised: same operations in the same order => identical bits
                np.testing.assert_array_equal(adv, g[k + "adv"], err_msg=k)
                np.testing.assert_array_equal(ret, g[k + "ret"], err_msg=k)
            else:
                # centralised: the row mean is summed in a different order than torch's
                ea, er = exact_centralised(g, k)
                print(k, "adv", check_centralised(adv, g[k + "adv"], ea, k + "adv"))
                print(k, "ret", check_centralised(ret, g[k + "ret"], er, k + "ret"))


@pytest.mark.parametrize("T,N", [(128, 32), (7, 1000), (128, 131072), (1, 1), (9, 257)])
@pytest.mark.parametrize("mode", [0, 1])
def test_gae_matches_oracle(oracle_mod, T, N, mode):
    from marl_amd.gae import compute_gae

    rng = np.random.default_rng(T * 31 + N)
    rewards = np.where(rng.random((T, N)) < 0.1, rng.random((T, N)), 0).astype(np.float32)
    dones = (rng.random((T, N)) < 0.02).astype(np.float32)
    values = rng.standard_normal((T, N)).astype(np.float32)
    nv = rng.standard_normal(1 if mode else N).astype(np.float32)
    adv_o, ret_o = oracle_mod.gae(rewards, dones, values, nv, 0.99, 0.95, mode)
    dev = "cuda:0"
    adv, ret = compute_gae(torch.from_numpy(rewards).to(dev), torch.from_numpy(dones).to(dev),
                           torch.from_numpy(values).to(dev), torch.from_numpy(nv).to(dev).reshape(1, -1), 0.99, 0.95,
                           centralised=bool(mode))
    if mode == 0:
        np.testing.assert_array_equal(adv.cpu().numpy(), adv_o)
        np.testing.assert_array_equal(ret.cpu().numpy(), ret_o)
    else:
        scale = np.abs(adv_o).max()
        np.testing.assert_allclose(adv.cpu().numpy(), adv_o, rtol=RTOL, atol=RTOL * scale)
        np.testing.assert_allclose(ret.cpu().numpy(), ret_o, rtol=RTOL, atol=RTOL * scale)


def test_row_sums():
    from marl_amd.gae import row_sums

    x = torch.rand(17, 100003, device="cuda:0")
    s = row_sums(x)
    torch.testing.assert_close(s, x.double().sum(dim=1), rtol=1e-12, atol=0)
