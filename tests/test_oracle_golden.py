"""Pin the oracle: it must reproduce the outputs of the unmodified reference stored in tests/golden/."""
import numpy as np
import pytest

from oracle import npg_oracle as O
from tests._cases import NPG_CASES, NpgCase, load
from oracle import synth


@pytest.mark.parametrize("name", NPG_CASES + ["npg_cfg4_small"])
def test_vpg_fvp_cg_match_reference(name):
    c = NpgCase(name)
    th = c.theta0.astype(np.float64)
    tr = c.transforms()
    a = (c.n, c.m, c.hidden)
    g = O.vpg(th, th, c.obs, c.act, c.adv_w, *a, tr, tr)
    c.check("vpg", g, 5e-6)
    gref = g.astype(np.float32) if c.big else c.g["vpg"]
    h = O.fvp(th, c.obs, gref.astype(np.float64), *a, tr, damping=1e-4)
    c.check("hvp_of_vpg", h, 5e-6)
    if not c.big:
        x = O.cg_solve(lambda p: O.fvp(th, c.obs, p, *a, tr, damping=1e-4), c.g["vpg"].astype(np.float64), c.cg_iters)
        c.check("cg_x", x, 2e-4)      # reference runs CG in fp32; fp64 truth sits ~1e-6..1e-4 away
    s = O.surrogate(th, th, c.obs, c.act, c.adv_w, *a, tr, tr)
    assert abs(s - float(c.g["surr_before"])) < 1e-6


@pytest.mark.parametrize("name", ["npg_cfg2_small", "npg_pointmass_32x32", "npg_cfg1_linear"])
def test_full_update_matches_reference(name):
    c = NpgCase(name)
    r = O.npg_update(c.theta0.astype(np.float64), c.obs, c.act, c.adv_w, c.n, c.m, c.hidden, c.transforms(),
                     cg_iters=c.cg_iters, damping=1e-4, delta=float(c.g["step"]))
    assert abs(r["alpha"] - float(c.g["alpha"])) / float(c.g["alpha"]) < 2e-4
    c.check("new_params", r["new_params"], 1e-4)
    assert abs(r["kl"] - float(c.g["kl"])) < 2e-4 * abs(float(c.g["kl"])) + 1e-7
    assert abs((r["surr_after"] - r["surr_before"]) - float(c.g["surr_improvement"])) < 1e-4


def test_trpo_line_search_matches_reference():
    c = NpgCase("trpo_cfg3_small")
    r = O.trpo_update(c.theta0.astype(np.float64), c.obs, c.act, c.adv_w, c.n, c.m, c.hidden, None,
                      cg_iters=c.cg_iters, kl_dist=float(c.g["kl_dist"]))
    assert r["tries"] == 2
    assert abs(r["alpha"] - float(c.g["alpha"])) / float(c.g["alpha"]) < 2e-4
    assert abs(r["kl"] - float(c.g["kl"])) < 1e-5


def test_torch_port_general_hvp_matches_reference():
    from oracle.torch_port import TorchPolicy
    g = load("hvp_general_64x64")
    n, m, hidden = int(g["n"]), int(g["m"]), tuple(int(h) for h in g["hidden"])
    paths = synth.make_paths(int(g["n_traj"]), int(g["T"]), n, m, seed=0)
    obs = np.concatenate([p["observations"] for p in paths]); act = np.concatenate([p["actions"] for p in paths])
    pol = TorchPolicy(g["theta_new"], n, m, hidden, theta_old=g["theta_old"],
                      tr_new=O.Transforms(n, m, g["in_shift"], g["in_scale"]))
    h = pol.hvp(obs, act, g["v"], 1e-4)
    assert np.linalg.norm(h - g["hvp"]) / np.linalg.norm(g["hvp"]) < 1e-5
    assert abs(float(pol.kl(obs, act)) - float(g["kl"])) < 1e-5


@pytest.mark.parametrize("kind", ["mlp", "quadratic", "linear"])
def test_gae_and_baselines_match_reference(kind):
    g = load("gae_" + kind)
    n, m = int(g["n"]), int(g["m"])
    paths = synth.make_paths(int(g["n_traj"]), int(g["T"]), n, m, seed=int(g["path_seed"]), ragged=True)
    gamma, lam = float(g["gamma"]), float(g["lam"])
    ret = np.concatenate([O.discount_sum(p["rewards"], gamma) for p in paths])
    np.testing.assert_allclose(ret, g["returns"], rtol=1e-13, atol=1e-13)
    obs_list = [p["observations"] for p in paths]
    if kind == "mlp":
        feat = O.mlp_baseline_features(obs_list).astype(np.float32)
        th = g["bl_params"]; sizes = [(128, n + 4), (128,), (128, 128), (128,), (1, 128), (1,)]
        parts, k = [], 0
        for s in sizes:
            cnt = int(np.prod(s)); parts.append(th[k:k + cnt].reshape(s)); k += cnt
        pred = O.mlp_baseline_forward(parts[0::2], parts[1::2], feat)
        np.testing.assert_allclose(pred, g["baseline_pred"], rtol=2e-5, atol=2e-6)
        pred = g["baseline_pred"].astype(np.float64)
    else:
        F = O.quadratic_baseline_features(obs_list) if kind == "quadratic" else O.linear_baseline_features(obs_list)
        coef = O.ridge_fit(F, ret, 1e-3 if kind == "quadratic" else 1e-5)
        pred = F.dot(coef)
        np.testing.assert_allclose(pred, g["baseline_pred"], rtol=1e-6, atol=1e-7)
        assert abs(np.sum((ret - pred) ** 2) / np.sum(ret ** 2) - float(g["err_after"])) < 1e-8
    adv, k = [], 0
    for p in paths:
        T = len(p["rewards"])
        adv.append(O.gae_path(p["rewards"], pred[k:k + T], p["terminated"], gamma, lam)); k += T
    np.testing.assert_allclose(np.concatenate(adv), g["advantages"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(ret - pred, g["advantages_nogae"], rtol=1e-9, atol=1e-9)
