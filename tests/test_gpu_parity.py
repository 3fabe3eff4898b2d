"""GPU parity tests: the HIP path (through the C ABI / ctypes) vs the golden fixtures produced by
the unmodified reference and vs the fp64 oracle.

Tolerances (relative L2 unless noted), from SURVEY section 6: the reference's own fp32 path sits
4.9e-7 (VPG) / 1e-7 (HVP) / 1.6e-6 (CG step direction) from fp64 truth; north-star bar is 1e-5
on the NPG step direction.
"""
import os

import numpy as np
import pytest

from oracle import npg_oracle as O
from oracle import synth
from tests._cases import NPG_CASES, NpgCase, load

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TOL_VPG = 3e-6
TOL_FVP = 3e-6
TOL_STEP = 1e-5          # the north-star bar
TOL_KL = 1e-5            # mean KL of an update against the reference (r06: was 1e-4)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def make_engine(c, layerwise=False):
    import torch  # noqa: F401
    from mjrl_amd.engine import UpdateEngine
    if layerwise:
        os.environ["MJX_FORCE_LAYERWISE"] = "1"
    try:
        eng = UpdateEngine(c.n, c.m, c.hidden)
    finally:
        os.environ.pop("MJX_FORCE_LAYERWISE", None)
    return eng


def packed_tr(c):
    if c.tr is None:
        return np.concatenate([np.zeros(c.n), np.ones(c.n), np.zeros(c.m), np.ones(c.m)]).astype(np.float32)
    return np.concatenate([np.float32(x).ravel() for x in c.tr])


@pytest.mark.parametrize("layerwise", [False, True])
@pytest.mark.parametrize("name", NPG_CASES)
def test_kernels_vs_reference(name, layerwise):
    import torch
    c = NpgCase(name)
    eng = make_engine(c, layerwise)
    if not layerwise and len(c.hidden) == 2 and max(c.hidden) <= 64:
        assert eng.fused, "fused kernel should serve this shape"
    tr = packed_tr(c)
    eng.set_policy(c.theta0, c.theta0, tr, tr)
    eng.set_batch(c.obs, c.act, c.adv_w)
    g, surr = eng.surr_vpg()
    assert rel(g.cpu().numpy(), c.g["vpg"]) < TOL_VPG
    assert abs(surr - float(c.g["surr_before"])) < 1e-6
    v = torch.from_numpy(c.g["vpg"]).to(eng.device)
    hv = eng.fvp(v).cpu().numpy() + np.float32(1e-4) * c.g["vpg"]
    assert rel(hv, c.g["hvp_of_vpg"]) < TOL_FVP
    x, gx = eng.cg_solve(v, c.cg_iters, 1e-4)
    assert rel(x.cpu().numpy(), c.g["cg_x"]) < TOL_STEP
    assert abs(gx - float(np.dot(c.g["vpg"].astype(np.float64), c.g["cg_x"]))) < 5e-5 * abs(gx)   # (a scalar of the solve; the parity bar is on x)
    eng.close()


@pytest.mark.parametrize("name", NPG_CASES)
def test_npg_agent_update_vs_reference(name):
    """NPG.train_from_paths through the mjrl-shaped classes == the reference's train_from_paths."""
    from mjrl_amd.algos.npg_cg import NPG
    from mjrl_amd.policies.gaussian_mlp import MLP, LinearPolicy
    c = NpgCase(name)
    spec = type("Spec", (), dict(observation_dim=c.n, action_dim=c.m, horizon=1000))
    pol = MLP(spec, hidden_sizes=c.hidden, seed=1, init_log_std=-0.5) if c.hidden else LinearPolicy(spec, seed=1, init_log_std=-0.5)
    pol.set_param_values(c.theta0)
    if c.tr is not None:
        pol.model.set_transformations(*c.tr)
        pol.old_model.set_transformations(*c.tr)
    agent = NPG(None, pol, None, normalized_step_size=float(c.g["step"]), FIM_invert_args={'iters': c.cg_iters, 'damping': 1e-4},
                save_logs=True)
    stats = agent.train_from_paths(c.paths)
    step, ref = pol.get_param_values().astype(np.float64) - c.theta0, c.g["new_params"].astype(np.float64) - c.theta0
    assert rel(step, ref) < TOL_STEP, rel(step, ref)
    lg = agent.logger.get_current_log()
    assert abs(lg["alpha"] - float(c.g["alpha"])) < 1e-5 * float(c.g["alpha"])
    assert abs(lg["kl_dist"] - float(c.g["kl"])) < TOL_KL * float(c.g["kl"]), ("kl", lg["kl_dist"], float(c.g["kl"]))
    assert abs(lg["surr_improvement"] - float(c.g["surr_improvement"])) < 2e-5
    np.testing.assert_allclose(stats, c.g["base_stats"], rtol=1e-12)
    assert pol.old_equals_new()


def test_trpo_line_search_vs_reference():
    from mjrl_amd.algos.trpo import TRPO
    from mjrl_amd.policies.gaussian_mlp import MLP
    c = NpgCase("trpo_cfg3_small")
    spec = type("Spec", (), dict(observation_dim=c.n, action_dim=c.m, horizon=1000))
    pol = MLP(spec, hidden_sizes=c.hidden, seed=1, init_log_std=-0.5)
    pol.set_param_values(c.theta0)
    agent = TRPO(None, pol, None, kl_dist=float(c.g["kl_dist"]), FIM_invert_args={'iters': c.cg_iters, 'damping': 1e-4})
    agent.train_from_paths(c.paths)
    assert agent.last_update["trials"] == 2                  # the fixture backtracks exactly once
    assert abs(agent.last_update["alpha"] - float(c.g["alpha"])) < 1e-5 * float(c.g["alpha"])
    assert abs(agent.last_update["kl_dist"] - float(c.g["kl"])) < TOL_KL * float(c.g["kl"]), ("kl", agent.last_update["kl_dist"], float(c.g["kl"]))
    step, ref = pol.get_param_values().astype(np.float64) - c.theta0, c.g["new_params"].astype(np.float64) - c.theta0
    assert rel(step, ref) < TOL_STEP


def test_big_net_layerwise_vs_reference():
    """cfg4 shapes (obs 376, act 17, 256x256) on the layer-wise path, N << d fixture: gradient and one Fisher product.
    (The step-level checks run on the well-conditioned N >= d fixture below.)"""
    import torch
    c = NpgCase("npg_cfg4_small")
    eng = make_engine(c)
    assert not eng.fused
    tr = packed_tr(c)
    eng.set_policy(c.theta0, c.theta0, tr, tr)
    eng.set_batch(c.obs, c.act, c.adv_w)
    g, _ = eng.surr_vpg()
    gh = g.cpu().numpy()
    c.check("vpg", gh, TOL_VPG)
    hv = eng.fvp(g).cpu().numpy() + np.float32(1e-4) * gh
    c.check("hvp_of_vpg", hv, TOL_STEP)          # (the product of OUR gradient, which sits ~1e-6 from the reference's)
    eng.close()


_WIDE = {}


def wide_case(name):
    """the N >= d fixtures regenerate 10^7..10^8 random numbers: build each once per session"""
    if name not in _WIDE:
        _WIDE[name] = NpgCase(name)
    return _WIDE[name]


def test_cfg4_wide_kernels_and_npg_update_vs_reference():
    """BASELINE configs[3] shapes (obs 376, act 17, 256x256, 25 CG iterations) at N = 200 000 >= d = 166 690
    (tests/golden/make_golden_big.py): gradient, Fisher product, the 25-iteration CG solve and the whole
    NPG.train_from_paths against the unmodified reference -- the step direction at the north-star bar."""
    from mjrl_amd.algos.npg_cg import NPG
    from mjrl_amd.policies.gaussian_mlp import MLP
    c = wide_case("npg_cfg4_wide")
    eng = make_engine(c)
    assert not eng.fused
    tr = packed_tr(c)
    eng.set_policy(c.theta0, c.theta0, tr, tr)
    eng.set_batch(c.obs, c.act, c.adv_w)
    g, surr = eng.surr_vpg()
    gh = g.cpu().numpy()
    c.check("vpg", gh, TOL_VPG)
    assert abs(surr - float(c.g["surr_before"])) < 1e-6
    hv = eng.fvp(g).cpu().numpy() + np.float32(1e-4) * gh
    c.check("hvp_of_vpg", hv, TOL_STEP)
    x, gx = eng.cg_solve(g, c.cg_iters, 1e-4)
    print("cfg4 wide cg_x:", c.check_step("cg_x", x.cpu().numpy(), TOL_STEP))
    eng.close()
    spec = type("Spec", (), dict(observation_dim=c.n, action_dim=c.m, horizon=1000))
    pol = MLP(spec, hidden_sizes=c.hidden, seed=1, init_log_std=-0.5)
    pol.set_param_values(c.theta0)
    agent = NPG(None, pol, None, normalized_step_size=float(c.g["step"]), FIM_invert_args={'iters': c.cg_iters, 'damping': 1e-4})
    stats = agent.train_from_paths(c.paths)
    step = pol.get_param_values().astype(np.float64) - c.theta0
    print("cfg4 wide update step:", c.check_step("update_step", step, TOL_STEP))
    assert abs(agent.last_update["alpha"] - float(c.g["alpha"])) < 1e-5 * float(c.g["alpha"])
    assert abs(agent.last_update["kl_dist"] - float(c.g["kl"])) < TOL_KL * float(c.g["kl"]), ("kl", agent.last_update["kl_dist"], float(c.g["kl"]))
    np.testing.assert_allclose(stats, c.g["base_stats"], rtol=1e-12)
    agent.engine.close()


def test_dapg_cfg5_wide_vs_reference():
    """BASELINE configs[4] shapes (obs 39, act 28, 512x512, DAPG with demonstrations, 10 CG iterations) at
    N = 300 000 on-policy timesteps >= d = 297 528: DAPG.train_from_paths against the unmodified reference
    (mjrl/algos/dapg.py:54-141), step direction at the north-star bar."""
    from mjrl_amd.algos.dapg import DAPG
    from mjrl_amd.policies.gaussian_mlp import MLP
    c = wide_case("dapg_cfg5_wide")
    spec = type("Spec", (), dict(observation_dim=c.n, action_dim=c.m, horizon=1000))
    pol = MLP(spec, hidden_sizes=c.hidden, seed=1, init_log_std=-0.5)
    pol.set_param_values(c.theta0)
    agent = DAPG(None, pol, None, demo_paths=c.demo_paths, kl_dist=float(c.g["kl_dist"]), lam_0=float(c.g["lam_0"]),
                 lam_1=float(c.g["lam_1"]), FIM_invert_args={'iters': c.cg_iters, 'damping': 1e-4})
    assert not agent.engine.fused
    stats = agent.train_from_paths(c.paths)
    step = pol.get_param_values().astype(np.float64) - c.theta0
    print("dapg wide update step:", c.check_step("update_step", step, TOL_STEP))
    assert abs(agent.last_update["alpha"] - float(c.g["alpha"])) < 1e-5 * float(c.g["alpha"])
    assert abs(agent.last_update["kl_dist"] - float(c.g["kl"])) < TOL_KL * float(c.g["kl"]), ("kl", agent.last_update["kl_dist"], float(c.g["kl"]))
    assert abs((agent.last_update["surr_after"] - agent.last_update["surr_before"]) - float(c.g["surr_improvement"])) < 2e-5
    np.testing.assert_allclose(stats, c.g["base_stats"], rtol=1e-12)
    agent.engine.close()


@pytest.mark.parametrize("key", ["configs3_humanoid_256x256", "configs4_adroit_512x512"])
def test_shard_size_update_vs_reference(key):
    """The sizes bench.py's layer-wise numbers are quoted on -- the per-GPU shards of BASELINE configs[3] / [4]: 500 000 x
    (376, 17, 256^2, NPG, 25 CG) and 1M x (39, 28, 512^2) + 5 000 demonstration rows (DAPG, 10 CG) -- through
    NPG / DAPG.train_from_paths on the path list, against the UNMODIFIED reference's run on the same seeded rows
    (tests/golden/make_golden_big.py npg_cfg4_shard / dapg_cfg5_shard; mjrl/algos/npg_cg.py:91-163, dapg.py:54-141)."""
    import bench
    from mjrl_amd.algos.dapg import DAPG
    from mjrl_amd.algos.npg_cg import NPG
    from mjrl_amd.policies.gaussian_mlp import MLP
    c = bench.LW_SHARDS[key]
    fx = os.path.join(ROOT, "tests", "golden", c["fixture"] + ".npz")
    if not os.path.exists(fx):
        pytest.skip("fixture %s not generated" % c["fixture"])
    g = np.load(fx)
    inp = bench.lw_shard_inputs(key)
    np.testing.assert_allclose(bench.lw_inputs_digest(inp), g["digest"], rtol=1e-12, atol=1e-9)      # the rows the reference saw
    n, m, hid, T = c["n"], c["m"], c["hidden"], c["T"]
    spec = type("Spec", (), dict(observation_dim=n, action_dim=m, horizon=T))
    pol = MLP(spec, hidden_sizes=hid, seed=1, init_log_std=-0.5)
    pol.set_param_values(inp["theta"])
    obs64, act64 = inp["obs"].astype(np.float64), inp["act"].astype(np.float64)
    paths = [dict(observations=obs64[i * T:(i + 1) * T], actions=act64[i * T:(i + 1) * T], rewards=np.zeros(T),
                  advantages=inp["adv"][i * T:(i + 1) * T].copy(), terminated=False) for i in range(c["n_traj"])]
    kw = dict(FIM_invert_args={'iters': c["cg_iters"], 'damping': 1e-4})
    if c["algo"] == "npg":
        agent = NPG(None, pol, None, normalized_step_size=0.05, **kw)
    else:
        Td = 200
        demos = [dict(observations=inp["demo_obs"][i * Td:(i + 1) * Td].astype(np.float64),
                      actions=inp["demo_act"][i * Td:(i + 1) * Td].astype(np.float64)) for i in range(c["demo_rows"] // Td)]
        agent = DAPG(None, pol, None, demo_paths=demos, kl_dist=c["kl_dist"], lam_0=c["lam_0"], lam_1=0.95, **kw)
    assert not agent.engine.fused
    agent.train_from_paths(paths)
    S = int(g["stride"])
    step = (pol.get_param_values().astype(np.float64) - inp["theta"])[::S]
    ref = g["update_step_sub"].astype(np.float64)
    err = float(np.linalg.norm(step - ref) / np.linalg.norm(ref))
    lu = agent.last_update
    print("%s: step %.2e from the reference (reference vs fp64 oracle: %s), alpha %.2e, kl %.2e"
          % (c["fixture"], err, ("%.2e" % float(g["err_ref_vs_f64_update_step"])) if "err_ref_vs_f64_update_step" in g.files else "n/a",
             abs(lu["alpha"] - float(g["alpha"])) / float(g["alpha"]), abs(lu["kl_dist"] - float(g["kl"])) / float(g["kl"])))
    assert err < TOL_STEP, err
    assert abs(lu["alpha"] - float(g["alpha"])) < 1e-5 * float(g["alpha"])
    assert abs(lu["kl_dist"] - float(g["kl"])) < TOL_KL * float(g["kl"]), ("kl", lu["kl_dist"], float(g["kl"]))
    assert abs((lu["surr_after"] - lu["surr_before"]) - float(g["surr_improvement"])) < 1e-4 * abs(float(g["surr_improvement"])) + 2e-6
    agent.engine.close()


@pytest.mark.parametrize("N", [1, 31, 32, 33, 1000, 4097])
def test_ragged_tails_vs_oracle(N):
    """N not a multiple of the 32-sample tile, N smaller than one tile, N == 1."""
    import torch
    from mjrl_amd.engine import UpdateEngine
    n, m, hid = 11, 3, (32, 32)
    rng = np.random.RandomState(N)
    th = synth.perturbed_params(synth.init_params(n, m, hid))
    obs, act, adv = rng.randn(N, n), rng.randn(N, m), rng.randn(N)
    eng = UpdateEngine(n, m, hid)
    tr = np.concatenate([np.zeros(n), np.ones(n), np.zeros(m), np.ones(m)]).astype(np.float32)
    eng.set_policy(th, th, tr, tr)
    eng.set_batch(obs, act, adv)
    th64 = th.astype(np.float64)
    g, surr = eng.surr_vpg()
    assert rel(g.cpu().numpy(), O.vpg(th64, th64, obs, act, adv, n, m, hid)) < TOL_VPG
    assert abs(surr - adv.mean()) < 1e-6
    v = rng.randn(th.size).astype(np.float32)
    hv = eng.fvp(torch.from_numpy(v).to(eng.device)).cpu().numpy()
    assert rel(hv, O.fvp(th64, obs, v.astype(np.float64), n, m, hid)) < TOL_FVP
    eng.close()


def test_old_neq_new_surrogate_kl_and_vpg():
    """K1 / K3 with theta_new != theta_old and different input transforms (the input_normalization
    situation, npg_cg.py:101-107) against the reference's own outputs."""
    from mjrl_amd.engine import UpdateEngine
    g = load("hvp_general_64x64")
    n, m, hid = int(g["n"]), int(g["m"]), tuple(int(h) for h in g["hidden"])
    paths = synth.make_paths(int(g["n_traj"]), int(g["T"]), n, m, seed=0)
    from tests._cases import fake_advantages
    fake_advantages(paths, 5)
    obs = np.concatenate([p["observations"] for p in paths]); act = np.concatenate([p["actions"] for p in paths])
    adv = O.whiten(np.concatenate([p["advantages"] for p in paths]))
    eng = UpdateEngine(n, m, hid)
    ident = np.concatenate([np.zeros(n), np.ones(n), np.zeros(m), np.ones(m)]).astype(np.float32)
    trn = np.concatenate([np.float32(g["in_shift"]), np.float32(g["in_scale"]), np.zeros(m, np.float32), np.ones(m, np.float32)])
    eng.set_policy(g["theta_new"], g["theta_old"], trn, ident)
    eng.set_batch(obs, act, adv)
    assert not eng.old_is_new
    surr, kl = eng.eval_surr_kl()
    assert abs(surr - float(g["surr"])) < 2e-6 and abs(kl - float(g["kl"])) < 1e-5 * float(g["kl"])
    gv, surr2 = eng.surr_vpg()
    # likelihood ratios span 7e-5 .. 1.3e3 here; the reference's own fp32 gradient sits 3.3e-6 from the
    # fp64 oracle on this case, so compare with the oracle tightly and with the reference at 1e-5
    tr64 = O.Transforms(n, m, g["in_shift"], g["in_scale"])
    truth = O.vpg(g["theta_new"].astype(np.float64), g["theta_old"].astype(np.float64), obs, act, adv, n, m, hid, tr64, None)
    assert rel(gv.cpu().numpy(), truth) < 5e-6
    assert rel(gv.cpu().numpy(), g["vpg"]) < 1e-5 and abs(surr2 - float(g["surr"])) < 2e-6
    eng.close()


def test_shard_sum_parity():
    """Multi-GPU math on one device: R trajectory shards bound one after the other with the global
    N; the sum of their partial gradients / Fisher-vector products equals the unsharded result."""
    import torch
    c = NpgCase("npg_cfg2_small")
    eng = make_engine(c)
    tr = packed_tr(c)
    eng.set_policy(c.theta0, c.theta0, tr, tr)
    eng.set_batch(c.obs, c.act, c.adv_w)
    g_full = eng.surr_vpg()[0].clone()
    v = g_full.clone()
    h_full = eng.fvp(v).clone()
    N = c.obs.shape[0]
    cuts = [0, 1500, 4000, 4001, N]
    g_sum, h_sum = torch.zeros_like(g_full), torch.zeros_like(h_full)
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        eng.set_batch(c.obs[lo:hi], c.act[lo:hi], c.adv_w[lo:hi], N_global=N)
        g_sum += eng.surr_vpg()[0]
        h_sum += eng.fvp(v)
    assert rel(g_sum.cpu().numpy(), g_full.cpu().numpy()) < 5e-7
    assert rel(h_sum.cpu().numpy(), h_full.cpu().numpy()) < 5e-7
    eng.close()


@pytest.mark.parametrize("transport,cut", [("peer", 23456), ("hook", 23456), ("peer", 0), ("peer3", 20000), ("peer-odd-d", 31000)])
def test_two_ranks_on_one_gpu_equal_one_rank(tmp_path, transport, cut):
    """The multi-rank control flow on real kernels: two processes (torch.distributed.run) share the GPU, each binds a ragged
    trajectory shard and runs the engine's update sequence (K1 + rank sum, the per-iteration FVP / rank sum / CG-step loop,
    device-side step length, K3 + rank sum, one read-back).  transport "peer": libmjx's peer exchange for real -- each process
    maps the other's buffer through HIP IPC, the Fisher product's reduction kernel stores into both and the stream waits on the
    arrival counter (no host synchronisation, no gloo in the data path); "hook": the same C loops with the sums handed to
    dist.all_reduce over gloo (RCCL refuses two ranks on one device).  cut = 0: rank 0 holds no trajectories at all -- it runs the
    exchange through the generic path (zeros into every buffer) while rank 1 runs it folded into its reduction / vector-update
    kernels.  Result == the one-process update on the whole batch (NPG call by call and as one call, TRPO with the device-side
    line search, DAPG as one call); all ranks hold bit-identical vectors.
    "peer3" (r04): THREE processes on the GPU -- the peer exchange with one arrival flag per source rank beyond two ranks (a 4-slot
    sum with one slot of zeros; ADVICE r03 asked for >= 3 ranks: this covers the protocol, not the ordering of real xGMI links).
    "peer-odd-d" (r06): the same 64 x 64 policy with FIVE actions -- d = 5 642 is not a multiple of 4, so the vector exchanges are
    NOT folded into the loop's kernels (generic push / sum launches) while the kernels still write accumulator-order partials.
    EIGHT processes (r06): tests/test_a_eight_ranks_gpu.py -- in a file of its own that runs FIRST, with the pytest process off the GPU."""
    import subprocess
    import sys
    import torch
    from mjrl_amd.engine import UpdateEngine
    out = str(tmp_path / "two_rank.npz")
    port = 29600 + (os.getpid() % 300)
    odd = transport == "peer-odd-d"
    if odd:
        transport = "peer"
    world = 3 if transport == "peer3" else 8 if transport == "peer8" else 2
    cuts3 = [20000, 41000] if world != 8 else [7000, 15000, 22000, 22000, 38000, 45000, 52500]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", MJX_PEER_COMM="1" if transport.startswith("peer") else "0",
               MJX_TEST_CUT=str(cut), MJX_TEST_CUTS=",".join(str(c) for c in cuts3), **({"MJX_TEST_SHAPE": "17,5,64,64"} if odd else {}))
    port += (7 if transport == "peer" else 0) + (13 if cut == 0 else 0) + (29 if world == 3 else 0) + (41 if world == 8 else 0) + (53 if odd else 0)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "_two_rank_gpu_worker.py"), out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280 if world < 8 else 900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    two = np.load(out)
    assert bool(two["ranks_identical"][0])
    assert two["native_comm"].all(), "the rank sums must run inside libmjx's C loops (mjx_cg_solve / mjx_npg_update)"
    assert str(two["comm_kind"][0]) == ("peer" if world > 2 else transport), (str(two["comm_kind"][0]), r.stderr[-3000:])
    assert bool(two["one_call_equal"][0]), "mjx_npg_update != the call-by-call sequence on two ranks"
    n, m, hid, N = (17, 5, (64, 64), 60000) if odd else (17, 6, (64, 64), 60000)
    rng = np.random.RandomState(5)
    obs, act, adv = rng.randn(N, n).astype(np.float32), rng.randn(N, m).astype(np.float32), rng.randn(N).astype(np.float32)
    th = synth.perturbed_params(synth.init_params(n, m, hid))
    assert (th.size % 4 != 0) == odd
    ident = np.concatenate([np.zeros(n), np.ones(n), np.zeros(m), np.ones(m)]).astype(np.float32)
    eng = UpdateEngine(n, m, hid)
    eng.set_policy(th, th, ident, ident)
    eng.set_batch(obs, act, adv)
    g, _ = eng.surr_vpg(sync=False)
    eng.cg_solve(g, 10, 1e-4, sync=False)
    eng.apply_npg_step(0.05, -3.0)
    surr_after, kl = eng.eval_surr_kl()
    late = eng.deferred()
    assert rel(two["grad"], g.cpu().numpy()) < 2e-6
    assert rel(two["x"], eng.x.cpu().numpy()) < TOL_STEP             # (CG amplifies the fp32 summation-order noise)
    assert rel(two["theta"], eng.theta_new.cpu().numpy()) < 1e-6
    one = np.array([late["surr_before"], late["gdotx"], late["alpha"], surr_after, kl])
    np.testing.assert_allclose(two["scal"], one, rtol=2e-5, atol=1e-7)
    # TRPO with the device-side line search: same number of trials, same step length / KL / parameters as on one rank
    eng.set_policy(th, th, ident, ident)
    tr = eng.trpo_update(10, 1e-4, 0.02, 0.002, -3.0)
    assert two["trpo"][4] == 1.0 and tr["accepted"] and int(two["trpo"][1]) == tr["trials"] and tr["trials"] > 3
    np.testing.assert_allclose(two["trpo"][[0, 2, 3]], [tr["alpha"], tr["kl"], tr["surr_after"]], rtol=2e-5, atol=1e-7)
    assert rel(two["trpo_theta"], eng.theta_new.cpu().numpy()) < 1e-6
    # DAPG: the ranks' blocks are [on-policy ; demonstrations] each; one rank sees the same rows as [all on-policy ; all demonstrations]
    bounds = [0, cut, N] if world == 2 else [0] + cuts3 + [N]
    on_idx, demo_idx = [], []
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        n_demo = min(500, hi - lo)
        on_idx += list(range(lo, hi - n_demo)); demo_idx += list(range(hi - n_demo, hi))
    idx = np.array(on_idx + demo_idx)
    n_on = len(on_idx)
    assert list(two["dapg_counts"]) == [n_on, N]
    adv_all = np.concatenate([adv[on_idx], 0.01 * np.ones(len(demo_idx), np.float32)])
    eng.set_policy(th, th, ident, ident)
    eng.set_batch(obs[idx], act[idx], adv_all)
    dres = eng.dapg_update(10, 1e-4, 0.05, -3.0, n_on, adv[on_idx])
    np.testing.assert_allclose(two["dapg"], list(dres), rtol=2e-5, atol=1e-7)
    assert rel(two["dapg_theta"], eng.theta_new.cpu().numpy()) < 1e-6
    eng.close()


def test_layerwise_block_beyond_2_31_elements():
    """no test had reached a block of >= 2^31 elements although `GemmArgs.M` and tile indices are `int`.  4 300 800
    rows x 512 hidden units = 2.2e9 floats per activation / tangent / delta block (8.8 GB each, ~40 GB of workspace): K1, a
    Fisher-vector product and a whole DAPG update (configs[4]'s architecture, 39-512-512-28) on the full block equal the sum /
    composition over two half-size shards whose blocks stay below 2^31 -- so no row, element or byte offset wraps.  Also the row
    limit of one layer-wise context (gridDim.y) is an error, not a wrap."""
    import ctypes
    import torch
    import bench
    free, total = torch.cuda.mem_get_info()
    if free < 120e9:
        pytest.skip("needs ~100 GB of free HBM (one 4.3M x 512^2 workspace + two half-size ones)")
    r = bench.full_size_case("configs4_adroit_512x512", rows=4300800, shards=2, time_it=False)
    assert r["rows"] * 512 > 2 ** 31
    # (product: a sum of positive semi-definite terms, two orders agree to 1e-7; gradient: an advantage-weighted sum of zero-mean terms
    #  in fp32 chains, measured 1.8e-6 between the two orders here -- a wrapped index would show as O(1), not O(1e-6))
    assert r["shard_sum_vs_full"]["gradient_rel_l2"] < 5e-6 and r["shard_sum_vs_full"]["fvp_rel_l2"] < 1e-6, r["shard_sum_vs_full"]
    u = r["update_vs_shard_composition"]
    assert u["alpha_rel"] < 1e-5 and u["kl_rel"] < 1e-4 and u["step_rel_l2"] < 3e-5, u
    assert not r["failed"]
    from mjrl_amd import _lib
    lib = _lib.load()
    ctx = ctypes.c_void_p()
    hid = (ctypes.c_int * 2)(512, 512)
    _lib.check(lib.mjx_create(ctypes.byref(ctx), 0, 39, 28, hid, 2))
    dummy = torch.zeros(64, device="cuda")
    assert lib.mjx_bind_batch(ctx, _lib.ptr(dummy), _lib.ptr(dummy), _lib.ptr(dummy), 65535 * 128 + 1, 65535 * 128 + 1) != 0
    assert b"at most" in lib.mjx_last_error()
    lib.mjx_destroy(ctx)


def test_bench_two_rank_path_on_one_gpu():
    """bench.py's N = 2 path (sharding, global whitening, barrier + max-over-ranks timing, rank-0 JSON line) with both
    ranks on this GPU and gloo in place of RCCL: same update as the N = 1 run (alpha, KL, surrogate improvement)."""
    import json
    import subprocess
    import sys
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-secondary"],
                         env=base, capture_output=True, text=True, timeout=280)
    assert one.returncode == 0, one.stderr[-3000:]
    j1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    # `python bench.py --gpus 2` on its own (WORLD_SIZE unset): the script launches its two ranks itself (the driver's command)
    env = dict(base, MJX_BENCH_SHARE_GPU="1", MJX_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=280)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-4000:]
    lines = [l for l in two.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                # rank 0 only
    j2 = json.loads(lines[0])
    assert j2["n_gpus"] == 2 and j2["scaling"] == "strong" and j2["value"] > 0 and j2["config"]["global_batch"] == j1["config"]["global_batch"]
    for k in ("alpha", "kl", "surr_improvement"):
        assert abs(j2["check"][k] - j1["check"][k]) <= 2e-4 * abs(j1["check"][k]) + 1e-9, (k, j1["check"], j2["check"])


@pytest.mark.parametrize("n,m,hid,N", [(39, 28, (32, 32), 20011), (11, 3, (64, 64), 20011)])
def test_whole_update_other_variants_vs_oracle(n, m, hid, N):
    """A whole NPG update (K1 with all three caches, 10 cached Fisher-vector products, device-side step, K3 through the
    old-policy cache) on other fused variants than cfg2's: the 32-action one at Adroit door-v0 sizes (its [tile][MP + 1]
    old-policy cache once overflowed an allocation sized for MP = 16) and the compile-time 12-feature instance."""
    from mjrl_amd.engine import UpdateEngine
    rng = np.random.RandomState(n + m)
    th = synth.perturbed_params(synth.init_params(n, m, hid), scale=0.05)
    obs, act, adv = rng.randn(N, n), rng.randn(N, m), rng.randn(N)
    adv = (adv - adv.mean()) / (adv.std() + 1e-6)
    tr = O.Transforms(n, m)
    pk = np.concatenate([tr.in_shift, tr.in_scale, tr.out_shift, tr.out_scale]).astype(np.float32)
    eng = UpdateEngine(n, m, hid)
    assert eng.fused
    for rep in range(2):                                   # (second pass: every cache is reused / revalidated)
        eng.set_policy(th, th, pk, pk)
        eng.set_batch(obs, act, adv)
        g, _ = eng.surr_vpg(sync=False)
        eng.cg_solve(g, 10, 1e-4, sync=False)
        eng.apply_npg_step(0.05, -3.0)
        surr_after, kl = eng.eval_surr_kl()
        late = eng.deferred()
    ref = O.npg_update(th.astype(np.float64), obs, act, adv, n, m, hid, tr, cg_iters=10, damping=1e-4, delta=0.05)
    assert rel(eng.x.cpu().numpy(), ref["npg"]) < TOL_STEP
    assert abs(late["alpha"] - ref["alpha"]) < 1e-4 * ref["alpha"]
    assert abs(kl - ref["kl"]) < 1e-4 * ref["kl"] + 1e-7
    assert abs(surr_after - ref["surr_after"]) < 2e-5 and abs(late["surr_before"] - ref["surr_before"]) < 2e-5
    eng.close()


@pytest.mark.parametrize("kind,n,N", [(2, 17, 100003), (2, 6, 777), (1, 39, 5000), (2, 11, 31), (2, 16, 4096), (1, 3, 1),
                                      (2, 39, 20011), (2, 24, 777), (2, 18, 33), (2, 45, 5)])      # > 176 features: 128 x 128 feature blocks
def test_gram_on_matrix_cores_equals_fma_gram(kind, n, N, monkeypatch):
    """mjx_bl_gram runs on the fp64 matrix cores (one workgroup holds all tiles up to 176 augmented features; 128 x 128 feature
    blocks beyond) and has an fp64 FMA register-tile kernel as the cross-check (MJX_GRAM_FMA=1).
    Same normal equations (to fp64 summation-order noise), symmetric, for ragged sample counts and partial tiles."""
    import torch
    from mjrl_amd import _lib
    from mjrl_amd._lib import check, ptr
    lib = _lib.load()
    rng = np.random.RandomState(n * 7 + N % 1000)
    obs = torch.from_numpy(rng.randn(N, n) * 4.0).cuda()          # (some entries beyond the +-10 clip)
    tpos = torch.from_numpy((np.arange(N) % 1000).astype(np.int32)).cuda()
    y = torch.from_numpy(rng.randn(N)).cuda()
    F = n + 5 if kind == 1 else n + n * (n + 1) // 2 + 5
    FA = F + 1
    out = []
    for fma in ("0", "1"):
        monkeypatch.setenv("MJX_GRAM_FMA", fma)
        G = torch.zeros(FA * FA, dtype=torch.float64).cuda()
        check(lib.mjx_bl_gram(kind, ptr(obs), ptr(tpos), ptr(y), N, n, ptr(G), None))
        torch.cuda.synchronize()
        out.append(G.cpu().numpy().reshape(FA, FA))
    a, b = out
    assert np.array_equal(a, a.T)
    assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max()


def test_fvp_properties_full_size():
    """BASELINE size (1M x 17, 64x64): size-independent properties of the Fisher-vector product --
    linearity, symmetry, positive semi-definiteness, and fused == layer-wise."""
    import torch
    from mjrl_amd.engine import UpdateEngine
    n, m, hid, N = 17, 6, (64, 64), 1000 * 1000
    rng = np.random.RandomState(0)
    obs = rng.randn(N, n).astype(np.float32)
    th = synth.perturbed_params(synth.init_params(n, m, hid))
    tr = np.concatenate([np.zeros(n), np.ones(n), np.zeros(m), np.ones(m)]).astype(np.float32)
    eng = UpdateEngine(n, m, hid)
    eng.set_policy(th, th, tr, tr)
    eng.set_batch(obs)
    v1 = torch.from_numpy(rng.randn(th.size).astype(np.float32)).to(eng.device)
    v2 = torch.from_numpy(rng.randn(th.size).astype(np.float32)).to(eng.device)
    h1, h2 = eng.fvp(v1).clone(), eng.fvp(v2).clone()
    h12 = eng.fvp(2.0 * v1 - 0.5 * v2).clone()
    assert rel(h12.cpu().numpy(), (2.0 * h1 - 0.5 * h2).cpu().numpy()) < 2e-6          # linearity
    a, b = float(torch.dot(v1.double(), h2.double())), float(torch.dot(v2.double(), h1.double()))
    assert abs(a - b) < 1e-5 * max(abs(a), abs(b), 1e-12)                               # symmetry
    assert float(torch.dot(v1.double(), h1.double())) > 0                                # PSD
    assert rel(eng.fvp(v1).cpu().numpy(), h1.cpu().numpy()) == 0.0                      # deterministic
    os.environ["MJX_FORCE_LAYERWISE"] = "1"
    try:
        lw = UpdateEngine(n, m, hid)
    finally:
        os.environ.pop("MJX_FORCE_LAYERWISE", None)
    lw.set_policy(th, th, tr, tr)
    lw.set_batch(obs)
    assert rel(lw.fvp(v1).cpu().numpy(), h1.cpu().numpy()) < 2e-6                       # two independent kernels agree
    # and a 20k-sample slice against the fp64 oracle
    eng.set_batch(obs[:20000])
    hv = eng.fvp(v1).cpu().numpy()
    assert rel(hv, O.fvp(th.astype(np.float64), obs[:20000].astype(np.float64), v1.cpu().numpy().astype(np.float64), n, m, hid)) < TOL_FVP
    eng.close(); lw.close()


@pytest.mark.parametrize("kind", ["mlp", "quadratic", "linear"])
def test_returns_and_gae_vs_reference(kind):
    """K5 against the reference's compute_returns / compute_advantages (ragged, terminated paths)."""
    from mjrl_amd.utils import process_samples
    g = load("gae_" + kind)
    n, m = int(g["n"]), int(g["m"])
    paths = synth.make_paths(int(g["n_traj"]), int(g["T"]), n, m, seed=int(g["path_seed"]), ragged=True)
    gamma, lam = float(g["gamma"]), float(g["lam"])
    process_samples.compute_returns(paths, gamma)
    np.testing.assert_allclose(np.concatenate([p["returns"] for p in paths]), g["returns"], rtol=1e-12, atol=1e-12)

    class Frozen:                      # baseline predictions taken from the fixture
        def __init__(self):
            self.k = 0
        def predict(self, path):
            T = len(path["rewards"]); out = g["baseline_pred"][self.k:self.k + T]; self.k += T
            return np.asarray(out, np.float64)
    process_samples.compute_advantages(paths, Frozen(), gamma, lam)
    np.testing.assert_allclose(np.concatenate([p["advantages"] for p in paths]), g["advantages"], rtol=1e-10, atol=1e-10)
    process_samples.compute_advantages(paths, Frozen(), gamma, None)
    np.testing.assert_allclose(np.concatenate([p["advantages"] for p in paths]), g["advantages_nogae"], rtol=1e-10, atol=1e-10)
    x = np.random.RandomState(1).randn(777)
    np.testing.assert_allclose(process_samples.discount_sum(x, 0.9), O.discount_sum(x, 0.9), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("N", [1, 30, 64, 127, 128, 200])
def test_mlp_baseline_fit_on_batches_smaller_than_two_minibatches(N):
    """MLPBaseline.fit runs int(N / 64) - 1 Adam steps per epoch (utils/optimize_model.py:24: the last, partial minibatch and one
    more are never visited): below 128 samples that is none -- parameters, moments and predictions stay as they were, the
    errors are finite and equal -- and from 128 on the parameters move."""
    from mjrl_amd.baselines.mlp_baseline import MLPBaseline
    spec = type("Spec", (), dict(observation_dim=5, action_dim=2, horizon=50))
    rng = np.random.RandomState(N)
    bl = MLPBaseline(spec, reg_coef=1e-3, batch_size=64, epochs=2, learn_rate=1e-3)
    paths = [dict(observations=rng.randn(N, 5), rewards=rng.randn(N), returns=rng.randn(N))]
    before = np.array(bl.params, copy=True)
    pred0 = np.array(bl.predict(paths[0]), copy=True)
    e0, e1 = bl.fit(paths, return_errors=True)
    assert np.isfinite(e0) and np.isfinite(e1) and np.all(np.isfinite(bl.params))
    if N < 128:
        assert np.array_equal(bl.params, before) and e0 == e1 and np.array_equal(bl.predict(paths[0]), pred0)
    else:
        assert not np.array_equal(bl.params, before)


def test_scans_on_edge_lengths_vs_oracle():
    """K5 on the trajectory lengths that sit on the scan kernel's seams -- one block of 256 threads per trajectory, the trajectory
    cut into 256 segments (csrc/vecops.h k_traj_scan): 1, 2, 255, 256, 257, 511, 512, 513, 1000, 4099 steps, terminated and
    not, in one batch -- against the sequential recurrences of the oracle (process_samples.py:21-44); > 64 KB of rewards, so the
    hand-out runs through the page-locked buffers as well."""
    from mjrl_amd.utils import process_samples
    rng = np.random.RandomState(11)
    lens = [1, 2, 255, 256, 257, 511, 512, 513, 1000, 4099, 1, 3, 4099, 256]
    paths = [dict(observations=rng.randn(T, 5), actions=rng.randn(T, 2), rewards=rng.randn(T), terminated=bool(i % 2)) for i, T in enumerate(lens)]
    gamma, lam = 0.995, 0.97
    process_samples.compute_returns(paths, gamma)
    for p in paths:
        np.testing.assert_allclose(p["returns"], O.discount_sum(p["rewards"], gamma), rtol=1e-12, atol=1e-12)
    base = [rng.randn(T) for T in lens]

    class Frozen:
        def __init__(self):
            self.k = 0
        def predict(self, path):
            self.k += 1
            return base[self.k - 1]
    process_samples.compute_advantages(paths, Frozen(), gamma, lam)
    for p, b in zip(paths, base):
        np.testing.assert_allclose(p["advantages"], O.gae_path(p["rewards"], b, p["terminated"], gamma, lam), rtol=1e-10, atol=1e-10)
    process_samples.compute_advantages(paths, Frozen(), gamma, None)
    for p, b in zip(paths, base):
        np.testing.assert_allclose(p["advantages"], p["returns"] - b, rtol=1e-12, atol=1e-12)


def test_vector_valued_rewards_and_baselines_vs_reference():
    """process_samples.py:26-27 (b.ndim == 2): (T, K) rewards against a (T, K) baseline -- K independent scans, column by column on
    the device -- against the UNMODIFIED reference's compute_returns / compute_advantages (oracle/ref_loader: sources or staged
    bytecode), GAE and plain, with and without normalisation, ragged and terminated paths."""
    import copy
    from mjrl_amd.utils import process_samples
    from oracle import ref_loader
    if ref_loader.install() is None:
        pytest.skip("reference not available")
    from mjrl.utils import process_samples as ref_ps
    rng = np.random.RandomState(4)
    K, lens = 3, [1, 7, 256, 300, 25]
    paths = [dict(observations=rng.randn(T, 5), actions=rng.randn(T, 2), rewards=rng.randn(T, K), terminated=bool(i % 2)) for i, T in enumerate(lens)]
    base = [rng.randn(T, K) for T in lens]

    class Frozen:
        def __init__(self):
            self.k = 0
        def predict(self, path):
            self.k += 1
            return base[self.k - 1]
    for lam, normalize in ((0.97, False), (0.97, True), (None, False), (None, True)):
        ours, ref = copy.deepcopy(paths), copy.deepcopy(paths)
        process_samples.compute_returns(ours, 0.99)
        ref_ps.compute_returns(ref, 0.99)
        process_samples.compute_advantages(ours, Frozen(), 0.99, lam, normalize=normalize)
        ref_ps.compute_advantages(ref, Frozen(), 0.99, lam, normalize=normalize)
        for a, b in zip(ours, ref):
            assert a["returns"].shape == b["returns"].shape and a["advantages"].shape == b["advantages"].shape == (len(a["rewards"]), K)
            np.testing.assert_allclose(a["returns"], b["returns"], rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(a["advantages"], b["advantages"], rtol=1e-10, atol=1e-10)
    # a 1-D reward vector against a (T, K) baseline broadcasts in the reference only when T == K: an error there, an error here
    bad = [dict(observations=rng.randn(9, 5), rewards=rng.randn(9), terminated=False)]
    process_samples.compute_returns(bad, 0.99)

    class Two:
        def predict(self, path):
            return np.zeros((9, 2))
    with pytest.raises(ValueError):
        process_samples.compute_advantages(bad, Two(), 0.99, 0.97)


def test_scan_full_size_properties():
    """1M timesteps: linearity of the scan and the one-step recurrence y[t] - g*y[t+1] == x[t]."""
    from mjrl_amd.utils import process_samples
    rng = np.random.RandomState(3)
    paths = [dict(rewards=rng.randn(1000)) for _ in range(1000)]
    process_samples.compute_returns(paths, 0.995)
    for p in paths[::97]:
        y, x = p["returns"], p["rewards"]
        np.testing.assert_allclose(y[:-1] - 0.995 * y[1:], x[:-1], rtol=0, atol=1e-12)
        assert y[-1] == x[-1]


# ----------------------------------------------------------------------------- K6 baselines
def _gae_paths(g):
    n, m = int(g["n"]), int(g["m"])
    paths = synth.make_paths(int(g["n_traj"]), int(g["T"]), n, m, seed=int(g["path_seed"]), ragged=True)
    k = 0
    for p in paths:
        T = len(p["rewards"]); p["returns"] = g["returns"][k:k + T].copy(); k += T
    return paths, n


@pytest.mark.parametrize("kind", ["quadratic", "linear"])
def test_ridge_baselines_vs_reference(kind):
    from mjrl_amd.baselines.quadratic_baseline import QuadraticBaseline
    from mjrl_amd.baselines.linear_baseline import LinearBaseline
    g = load("gae_" + kind)
    paths, n = _gae_paths(g)
    spec = type("Spec", (), dict(observation_dim=n, action_dim=3, horizon=400))
    bl = (QuadraticBaseline if kind == "quadratic" else LinearBaseline)(spec)
    assert np.all(bl.predict(paths[0]) == 0.0)                     # unfitted: zeros (quadratic_baseline.py:72-73)
    e0, e1 = bl.fit(paths, return_errors=True)
    assert e0 == 1.0 and abs(e1 - float(g["err_after"])) < 1e-9
    pred = bl.predict_batch(paths)
    np.testing.assert_allclose(pred, g["baseline_pred"], rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(bl.predict(paths[3]), pred[sum(len(p["rewards"]) for p in paths[:3]):][:len(paths[3]["rewards"])], rtol=0, atol=1e-12)
    # the Gram matrix itself against the fp64 feature matrix of the oracle
    from mjrl_amd.baselines._features import DeviceBlock, FEAT_QUADRATIC, FEAT_LINEAR
    obs_list = [p["observations"] for p in paths]
    F = O.quadratic_baseline_features(obs_list) if kind == "quadratic" else O.linear_baseline_features(obs_list)
    G = DeviceBlock(paths, 'obs').gram(FEAT_QUADRATIC if kind == "quadratic" else FEAT_LINEAR, g["returns"])
    Fa = np.concatenate([F, g["returns"][:, None]], axis=1)
    np.testing.assert_allclose(G, Fa.T @ Fa, rtol=1e-11, atol=1e-9)
    assert np.array_equal(G, G.T)


def test_mlp_baseline_vs_reference():
    import torch
    from mjrl_amd.baselines.mlp_baseline import MLPBaseline
    g = load("gae_mlp")
    paths, n = _gae_paths(g)
    spec = type("Spec", (), dict(observation_dim=n, action_dim=3, horizon=400))
    torch.manual_seed(4); np.random.seed(4)
    bl = MLPBaseline(spec, reg_coef=1e-3, batch_size=64, epochs=2, learn_rate=1e-3)
    assert np.array_equal(bl.params, g["bl_params"])               # same seed -> the reference's initial weights
    pred = bl.predict_batch(paths)
    np.testing.assert_allclose(pred, g["baseline_pred"], rtol=2e-5, atol=2e-6)
    np.random.seed(int(g["fit_seed"]))
    e0, e1 = bl.fit(paths, return_errors=True)
    assert abs(e0 - float(g["err_before"])) < 1e-6
    # 2 epochs x 94 Adam steps with identical minibatches: fp32 round-off only
    assert abs(e1 - float(g["err_after"])) < 2e-4 * float(g["err_after"])
    assert rel(bl.params, g["bl_params_after"]) < 2e-4
    assert bl.adam_steps == 2 * (len(g["returns"]) // 64 - 1)
    import pickle
    bl2 = pickle.loads(pickle.dumps(bl))
    np.testing.assert_array_equal(bl2.predict(paths[0]), bl.predict(paths[0]))


@pytest.mark.parametrize("n", [60, 111, 376])
def test_wide_mlp_baseline_vs_the_reference_run_live(n):
    """MLPBaseline at observation widths beyond one workgroup's LDS (Ant 111, Humanoid 376 = BASELINE configs[3]; 60 + 4 = two
    workgroups): the several-workgroup persistent trainer (csrc/mlp_fit.h, MULTI) through the host class against the UNMODIFIED
    reference's MLPBaseline (mjrl/baselines/mlp_baseline.py:61-105, utils/optimize_model.py:7-36; oracle/ref_loader: sources or staged
    bytecode) run here on the CPU with the same seeds -- same initial weights, same predictions, and after 2 epochs x 39 Adam steps
    on the same permutations the same parameters to fp32 round-off."""
    import torch
    from mjrl_amd.baselines.mlp_baseline import MLPBaseline
    from mjrl_amd.utils import process_samples
    from oracle import ref_loader
    if ref_loader.install() is None:
        pytest.skip("reference not available")
    from mjrl.baselines.mlp_baseline import MLPBaseline as RefBaseline
    from mjrl.utils import process_samples as ref_ps
    rng = np.random.RandomState(n)
    w = rng.randn(n) / np.sqrt(n)
    paths = []
    for _ in range(20):
        obs = np.cumsum(0.1 * rng.randn(128, n), axis=0) + rng.randn(n)
        paths.append(dict(observations=obs, rewards=np.tanh(obs @ w) + 0.1 * rng.randn(128), terminated=False))
    ref_paths = [dict(p) for p in paths]
    ref_ps.compute_returns(ref_paths, 0.995)
    process_samples.compute_returns(paths, 0.995)
    spec = type("Spec", (), dict(observation_dim=n, action_dim=3, horizon=128))
    kw = dict(reg_coef=1e-3, batch_size=64, epochs=2, learn_rate=1e-3)
    flat_of = lambda b: np.concatenate([p.data.numpy().ravel() for p in b.model.parameters()])
    # the yardstick: how far the REFERENCE lands from itself when its initial weights move by 1e-7 (a ReLU / Adam chain is stable
    # on most data -- 1e-7 -> 1e-7 -- and chaotic on some: the width-111 instance of this test ends 1.3e-3 apart, a unit's sign
    # flips and Adam's normalised steps amplify it)
    torch.manual_seed(4); np.random.seed(4)
    twin = RefBaseline(spec, **kw)
    gen = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for p_ in twin.model.parameters():
            p_.mul_(1 + 1e-7 * torch.randn(p_.shape, generator=gen))
    np.random.seed(11)
    twin.fit([dict(p) for p in ref_paths])
    torch.manual_seed(4); np.random.seed(4)
    ref = RefBaseline(spec, **kw)
    torch.manual_seed(4); np.random.seed(4)
    bl = MLPBaseline(spec, **kw)
    ref_flat = lambda: flat_of(ref)
    assert np.array_equal(bl.params, ref_flat())
    np.testing.assert_allclose(bl.predict_batch(paths), np.concatenate([ref.predict(p) for p in ref_paths]), rtol=2e-5, atol=2e-6)
    np.random.seed(11)
    r0, r1 = ref.fit(ref_paths, return_errors=True)
    np.random.seed(11)
    e0, e1 = bl.fit(paths, return_errors=True)
    sens = rel(flat_of(twin), ref_flat())
    print("width %d: parameters %.2e from the reference's after 78 Adam steps (the reference from itself under a 1e-7 perturbation: %.2e)"
          % (n, rel(bl.params, ref_flat()), sens))
    assert abs(e0 - r0) < 2e-6 * max(1.0, abs(r0)) and abs(e1 - r1) < (5e-4 + 10 * sens) * abs(r1), (e0, r0, e1, r1)
    assert rel(bl.params, ref_flat()) < max(2e-4, 10 * sens), (rel(bl.params, ref_flat()), sens)
    assert bl.adam_steps == 2 * (20 * 128 // 64 - 1)
    np.testing.assert_allclose(bl.predict(paths[0]), ref.predict(ref_paths[0]), rtol=2e-3 + 10 * sens, atol=2e-3 + 10 * sens)


def test_mlp_baseline_fit_quality_at_iteration_scale_vs_reference():
    """The persistent minibatch-Adam trainer against the UNMODIFIED reference at iteration scale (300 000 timesteps, 2 epochs =
    9 372 Adam steps on the same NumPy permutations; tests/golden/make_golden_mlpfit.py): after thousands of chaotic ReLU / Adam
    steps the parameters of two correct fp32 implementations differ, the QUALITY of the fit does not -- the reference's own
    error measures (mlp_baseline.py:83,94) and the mean squared error on held-out paths agree to a few per cent.  (Bit-level
    parity of the same trainer on short chains: test_mlp_baseline_vs_reference.)"""
    import torch
    from mjrl_amd.baselines.mlp_baseline import MLPBaseline
    from mjrl_amd.utils import process_samples
    g = load("mlpfit_300k")
    n, n_traj, T = int(g["n"]), int(g["n_traj"]), int(g["T"])

    def make_paths(k, seed):
        rng = np.random.RandomState(seed)
        w = np.random.RandomState(1234).randn(n) / np.sqrt(n)
        paths = []
        for _ in range(k):
            obs = np.cumsum(0.1 * rng.randn(T, n), axis=0) + rng.randn(n)
            rew = np.tanh(obs @ w) - 0.05 * np.sum(obs[:, :3] ** 2, axis=1) + 0.1 * rng.randn(T)
            paths.append(dict(observations=obs, rewards=rew, terminated=False))
        process_samples.compute_returns(paths, 0.995)
        return paths
    paths, held = make_paths(n_traj, int(g["path_seed"])), make_paths(20, int(g["held_seed"]))
    y = np.concatenate([p["returns"] for p in held])
    assert abs(y.mean() - float(g["ret_mean"])) < 1e-9 * abs(float(g["ret_mean"]))          # the same data as the reference saw
    spec = type("Spec", (), dict(observation_dim=n, action_dim=6, horizon=T))
    torch.manual_seed(int(g["init_seed"])); np.random.seed(int(g["init_seed"]))
    bl = MLPBaseline(spec, reg_coef=1e-3, batch_size=64, epochs=2, learn_rate=1e-3)
    np.random.seed(int(g["fit_seed"]))
    e0, e1 = bl.fit(paths, return_errors=True)
    assert bl.adam_steps == int(g["steps"])
    pred = np.concatenate([np.asarray(bl.predict(p)) for p in held])
    mse = float(np.mean((pred - y) ** 2))
    print("mlp fit at scale: err_before %.6f (ref %.6f)  err_after %.5f (ref %.5f)  held-out mse %.1f (ref %.1f)  pred mean %.2f (ref %.2f)"
          % (e0, float(g["err_before"]), e1, float(g["err_after"]), mse, float(g["held_mse"]), pred.mean(), float(g["pred_mean"])))
    assert abs(e0 - float(g["err_before"])) < 1e-5
    assert abs(e1 - float(g["err_after"])) < 0.02 * float(g["err_after"])            # measured 0.34 %
    assert abs(mse - float(g["held_mse"])) < 0.02 * float(g["held_mse"])            # measured 0.23 %
    assert abs(pred.std() - float(g["pred_std"])) < 0.02 * float(g["pred_std"])


def test_train_step_plumbing_numpy_env():
    """cfg1-style end-to-end train_step: NumPy point-mass stand-in env -> sampler -> returns/GAE ->
    NPG update -> baseline fit, all through the mjrl-shaped classes."""
    from mjrl_amd.algos.npg_cg import NPG
    from mjrl_amd.baselines.mlp_baseline import MLPBaseline
    from mjrl_amd.policies.gaussian_mlp import MLP

    class PointMass:                       # obs = [pos(2), vel(2), target(2)], act = force(2), horizon 25
        horizon = 25
        def __init__(self):
            self.rng = np.random.RandomState(0)
        def set_seed(self, s):
            self.rng = np.random.RandomState(s)
        def reset(self):
            self.p, self.v, self.g, self.t = self.rng.uniform(-1, 1, 2), np.zeros(2), self.rng.uniform(-1, 1, 2), 0
            return np.concatenate([self.p, self.v, self.g])
        def step(self, a):
            self.v = 0.9 * self.v + 0.1 * np.clip(a, -1, 1); self.p = self.p + 0.1 * self.v; self.t += 1
            return np.concatenate([self.p, self.v, self.g]), -float(np.linalg.norm(self.p - self.g)), False, {}

    spec = type("Spec", (), dict(observation_dim=6, action_dim=2, horizon=25))
    pol = MLP(spec, hidden_sizes=(32, 32), seed=2, init_log_std=-0.5)
    bl = MLPBaseline(spec, reg_coef=1e-3, batch_size=64, epochs=2, learn_rate=1e-3)
    agent = NPG(PointMass(), pol, bl, normalized_step_size=0.05, seed=2, save_logs=True)
    scores = []
    for _ in range(8):
        stats = agent.train_step(N=80, sample_mode='trajectories', gamma=0.95, gae_lambda=0.97, num_cpu=1)
        scores.append(stats[0]); assert len(stats) == 5 and np.isfinite(stats[0])
    lg = agent.logger.get_current_log()
    for k in ("alpha", "delta", "time_vpg", "time_npg", "kl_dist", "surr_improvement", "running_score", "num_samples",
              "time_VF", "VF_error_before", "VF_error_after", "stoc_pol_mean", "time_sampling"):
        assert k in lg, k
    assert 0 < lg["kl_dist"] < 0.1 and lg["surr_improvement"] > 0
    assert scores[-1] > scores[0]          # it learns
    assert agent.seed == 2 + 8 * 80


# ----------------------------------------------------------------------------- N1: general Hessian
def test_general_hvp_vs_reference():
    """theta_new != theta_old and different input transforms: exact Pearlmutter product against the
    reference's double-backward HVP (golden) and the torch-autograd port."""
    import torch
    from mjrl_amd.engine import UpdateEngine
    g = load("hvp_general_64x64")
    n, m, hid = int(g["n"]), int(g["m"]), tuple(int(h) for h in g["hidden"])
    paths = synth.make_paths(int(g["n_traj"]), int(g["T"]), n, m, seed=0)
    obs = np.concatenate([p["observations"] for p in paths])
    eng = UpdateEngine(n, m, hid)
    ident = np.concatenate([np.zeros(n), np.ones(n), np.zeros(m), np.ones(m)]).astype(np.float32)
    trn = np.concatenate([np.float32(g["in_shift"]), np.float32(g["in_scale"]), np.zeros(m, np.float32), np.ones(m, np.float32)])
    eng.set_policy(g["theta_new"], g["theta_old"], trn, ident)
    eng.set_batch(obs)
    hv = eng.fvp(torch.from_numpy(g["v"]).to(eng.device)).cpu().numpy() + np.float32(1e-4) * g["v"]
    assert rel(hv, g["hvp"]) < 1e-5, rel(hv, g["hvp"])
    eng.close()
    # a non-fused shape with out_scale != 1, against the autograd port
    from oracle.torch_port import TorchPolicy
    n, m, hid, N = 40, 9, (96, 48, 24), 3000
    rng = np.random.RandomState(2)
    th_o = synth.perturbed_params(synth.init_params(n, m, hid), scale=0.05)
    th_n = (th_o + 0.03 * rng.randn(th_o.size)).astype(np.float32)
    obs = rng.randn(N, n)
    act = rng.randn(N, m)
    trn = O.Transforms(n, m, 0.1 * rng.randn(n), 1 + 0.1 * rng.rand(n), 0.05 * rng.randn(m), 1 + 0.2 * rng.rand(m))
    tro = O.Transforms(n, m, None, None, 0.02 * rng.randn(m), 1 + 0.1 * rng.rand(m))
    pk = lambda t: np.concatenate([t.in_shift, t.in_scale, t.out_shift, t.out_scale]).astype(np.float32)
    v = rng.randn(th_o.size).astype(np.float32)
    ref = TorchPolicy(th_n, n, m, hid, theta_old=th_o, tr_new=trn, tr_old=tro).hvp(obs, act, v, 0.0)
    eng = UpdateEngine(n, m, hid)
    eng.set_policy(th_n, th_o, pk(trn), pk(tro))
    eng.set_batch(obs)
    hv = eng.fvp(torch.from_numpy(v).to(eng.device)).cpu().numpy()
    assert rel(hv, ref) < 1e-5, rel(hv, ref)
    eng.close()


def test_npg_input_normalization_vs_reference():
    from mjrl_amd.algos.npg_cg import NPG
    from mjrl_amd.policies.gaussian_mlp import MLP
    c = NpgCase("npg_inputnorm_32x32")
    spec = type("Spec", (), dict(observation_dim=c.n, action_dim=c.m, horizon=1000))
    pol = MLP(spec, hidden_sizes=c.hidden, seed=1, init_log_std=-0.5)
    pol.set_param_values(c.theta0)
    agent = NPG(None, pol, None, normalized_step_size=float(c.g["step"]), input_normalization=float(c.g["input_normalization"]),
                FIM_invert_args={'iters': c.cg_iters, 'damping': 1e-4})
    agent.train_from_paths(c.paths)
    np.testing.assert_allclose(pol.model.in_shift, c.g["final_in_shift"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(pol.model.in_scale, c.g["final_in_scale"], rtol=1e-6, atol=1e-7)
    assert np.all(pol.old_model.in_scale == 1.0)                  # the reference never touches old_model here
    step, ref = pol.get_param_values().astype(np.float64) - c.theta0, c.g["new_params"].astype(np.float64) - c.theta0
    # the bar: TOL_STEP to the reference; the general-Hessian solve differences mu_new - mu_old in fp32 (so does the
    # reference), so where two fp32 implementations land farther apart the bar is the distance to the fp64 truth
    # (npg_inputnorm_32x32_f64.npz, make_golden_big.py: the reference sits 2.0e-6 from it)
    t = load("npg_inputnorm_32x32_f64")
    e_ref, e_f64, ref_f64 = rel(step, ref), rel(step, t["update_step_f64"]), float(t["err_ref_vs_f64_update_step"])
    print("input_normalization step: vs reference %.2e, vs fp64 %.2e (reference vs fp64 %.2e)" % (e_ref, e_f64, ref_f64))
    assert e_ref < TOL_STEP, (e_ref, e_f64, ref_f64)
    assert abs(agent.last_update["alpha"] - float(c.g["alpha"])) < 1e-5 * float(c.g["alpha"])
    # (the general-position update: the reference's KL here is an fp32 mean over 3 000 samples of a policy with a freshly shifted input
    #  transform -- measured 1.24e-5 from ours, the one fixture above the 1e-5 the other updates' KL is held to)
    assert abs(agent.last_update["kl_dist"] - float(c.g["kl"])) < 2e-5 * float(c.g["kl"]), ("kl", agent.last_update["kl_dist"], float(c.g["kl"]))


def test_hvp_sample_frac_rng_parity():
    """hvp_sample_frac < 0.99: one with-replacement row sample per Fisher product from NumPy's global RNG
    (npg_cg.py:65-69); with the same seed the oracle replays the same rows."""
    from mjrl_amd.algos.npg_cg import NPG
    from mjrl_amd.policies.gaussian_mlp import MLP
    c = NpgCase("npg_cfg2_small")           # 10 000 timesteps, d = 5 708: the half-sample Fisher still has N ~ d rows
    spec = type("Spec", (), dict(observation_dim=c.n, action_dim=c.m, horizon=1000))
    pol = MLP(spec, hidden_sizes=c.hidden, seed=1, init_log_std=-0.5)
    pol.set_param_values(c.theta0)
    agent = NPG(None, pol, None, normalized_step_size=0.05, hvp_sample_frac=0.5, FIM_invert_args={'iters': 6, 'damping': 1e-4})
    np.random.seed(11)
    agent.train_from_paths(c.paths)
    th = c.theta0.astype(np.float64)
    a = (c.n, c.m, c.hidden)
    g = O.vpg(th, th, c.obs, c.act, c.adv_w, *a)
    np.random.seed(11)
    N = c.obs.shape[0]
    def hv(p):
        idx = np.random.choice(N, size=int(0.5 * N))
        return O.fvp(th, c.obs[idx], p, *a, damping=1e-4)
    x = O.cg_solve(hv, g, 6)
    alpha = np.sqrt(abs(0.05 / (g.dot(x) + 1e-20)))
    step = pol.get_param_values().astype(np.float64) - c.theta0
    assert rel(step, alpha * x) < TOL_STEP, rel(step, alpha * x)


@pytest.mark.parametrize("n,m,hid,expect_fused", [
    (11, 3, (64, 64), True),       # compile-time feature count 12 (Hopper-sized observations)
    (8, 2, (64, 64), True),        # ... 12 at its lower edge (n + 1 = 9)
    (6, 2, (64, 64), True),        # compile-time feature count 8
    (4, 1, (64, 64), True),        # ... and n + 1 = 5
    (19, 6, (64, 64), True),       # compile-time feature count 20 at its upper edge (no pad column)
    (17, 12, (64, 64), True),      # MP = 16 variant
    (9, 10, (32, 32), True),       # 32x32, MP = 16
    (40, 4, (32, 32), True),       # NT1 = 2 (obs dim > 31)
    (22, 8, (64, 64), True),       # m == MP
    (28, 8, (64, 64), None),       # near the 160 KB LDS limit of the 64x64 variant: either path
    (31, 1, (32, 32), True),       # n + 1 == 32 exactly, single action
    (12, 3, (64, 32), False),      # unequal hidden sizes -> layer-wise
    (50, 5, (64, 64), False),      # obs too wide for the 64x64 fused LDS budget -> layer-wise
    (8, 2, (48,), False),          # one hidden layer
    (39, 28, (32, 32), True),      # Adroit door-sized: the 32-action variant
    (46, 26, (32, 32), True),      # hammer-sized
    (7, 20, (32, 32), True),       # few observations, more than 16 actions: also the 32-action variant
    (7, 33, (32, 32), False),      # more actions than any fused variant
    (23, 3, (64, 128), False),     # layer-wise with the one-pass output layer (csrc/lw_head.h): last hidden layer 128 ...
    (17, 17, (256, 256), False),   # ... 256 (configs[3] widths) ...
    (12, 32, (128, 384), False),   # ... 384, 32 actions (its upper limit) ...
    (39, 28, (512, 512), False),   # ... 512 (configs[4]); N = 3000 + n leaves a partial 64-row tile in every case
    (10, 4, (256, 192), False),    # last hidden layer not a multiple of 128: the generic chain
    (40, 5, (256, 512, 256), False),   # three hidden layers through every r02 path: padded observation rows (40 -> 64), persistent
                                       # forward / tangent / delta products with one and two 256-column blocks, the eight-wave output pass
    (33, 2, (128,), False),        # one hidden layer of 128 units, 33 observations (-> 64): no fused output pass (needs two layers)
])
def test_other_shapes_vs_oracle(n, m, hid, expect_fused):
    """every kernel variant / dispatch branch: K1, K2, K3 against the fp64 oracle (with transforms, old != new in K3)"""
    import torch
    from mjrl_amd.engine import UpdateEngine
    rng = np.random.RandomState(n * 100 + m)
    N = 3000 + n
    th = synth.perturbed_params(synth.init_params(n, m, hid), scale=0.05)
    # (the second parameter set: per-weight noise scaled down with the layer width, so that the likelihood ratios of the wide
    #  nets stay O(1) -- at 0.02 per weight a 512-wide net moves every mean by ~0.5 sigma and exp(LL_new - LL_old) spans e^+-10)
    th2 = (th + 0.02 * (64.0 / max([64] + list(hid))) * rng.randn(th.size)).astype(np.float32)
    obs, act, adv = rng.randn(N, n), rng.randn(N, m), rng.randn(N)
    tr = O.Transforms(n, m, 0.1 * rng.randn(n), 1 + 0.1 * rng.rand(n), 0.05 * rng.randn(m), 1 + 0.2 * rng.rand(m))
    pk = np.concatenate([tr.in_shift, tr.in_scale, tr.out_shift, tr.out_scale]).astype(np.float32)
    eng = UpdateEngine(n, m, hid)
    assert expect_fused is None or eng.fused == expect_fused
    eng.set_policy(th, th, pk, pk)
    eng.set_batch(obs, act, adv)
    th64 = th.astype(np.float64)
    g, surr = eng.surr_vpg()
    assert rel(g.cpu().numpy(), O.vpg(th64, th64, obs, act, adv, n, m, hid, tr, tr)) < TOL_VPG
    v = rng.randn(th.size).astype(np.float32)
    hv = eng.fvp(torch.from_numpy(v).to(eng.device)).cpu().numpy()
    assert rel(hv, O.fvp(th64, obs, v.astype(np.float64), n, m, hid, tr)) < TOL_FVP
    eng.set_policy(th2, th, pk, pk)
    s, kl = eng.eval_surr_kl()
    t2 = th2.astype(np.float64)
    assert abs(s - O.surrogate(t2, th64, obs, act, adv, n, m, hid, tr, tr)) < 5e-6
    klo = O.mean_kl(t2, th64, obs, n, m, hid, tr, tr)
    assert abs(kl - klo) < 2e-5 * klo + 1e-7
    g2, s2 = eng.surr_vpg()                                   # K1 with an explicit old network
    # (old != new: every term carries LR = exp(LL_new - LL_old); with ~30 actions |LL| ~ 40, so fp32 rounding of the
    #  log-likelihoods alone is ~40 x 6e-8 relative in LR)
    #  (512-wide layers add the rounding of 512-term fp32 dot products in every mean: 1.3e-5 measured at 39-512-512-28)
    assert rel(g2.cpu().numpy(), O.vpg(t2, th64, obs, act, adv, n, m, hid, tr, tr)) < (5e-6 if m <= 16 else TOL_STEP if max(hid) <= 256 else 2e-5)
    eng.close()


@pytest.mark.parametrize("n,m,h1,h2,N", [(64, 6, 256, 256, 3000 + 37), (96, 28, 512, 512, 2 * 128 + 1), (128, 3, 256, 512, 1024),
                                         (88, 5, 256, 256, 1500 + 3)])      # (n = 88: observation rows padded to 96 = three whole k-tiles, like configs[3]'s 376 -> 384)
def test_persistent_gemm_bitwise_equals_general_kernel(tmp_path, n, m, h1, h2, N):
    """The persistent interior-tile GEMM (csrc/lw_gemm_p.h: tangent and delta products of the layer-wise Fisher-vector
    product) performs the same operations in the same order as the general kernel: the products must agree BIT FOR BIT
    with MJX_LW_PERSIST=0 -- including a batch that ends in a partial 128-row tile (padding rows computed, never summed)."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    res = {}
    for tag, val in (("general", "0"), ("persistent", "1")):
        env = dict(os.environ, MJX_LW_PERSIST=val)
        out = str(tmp_path / (tag + ".npz"))
        subprocess.run([sys.executable, os.path.join(here, "_lw_fvp_worker.py"), out, str(n), str(m), str(h1), str(h2), str(N)],
                       check=True, env=env, timeout=600)
        res[tag] = np.load(out)
    # weights: bit for bit.  Bias gradients of the hidden layers are the column sums formed in the delta product's epilogue:
    # the compiler contracts `sum += acc * factor` into an FMA in one kernel and not in the other (the row mask sits in
    # between) -- last-bit differences, so those blocks are compared at 1e-6 of the block's largest entry.
    bias = np.zeros(res["general"]["g"].size, bool)
    o = n * h1
    bias[o:o + h1] = True
    o += h1 + h1 * h2
    bias[o:o + h2] = True
    for k in ("g", "hv"):
        a, b = res["general"][k], res["persistent"][k]
        assert np.array_equal(a[~bias], b[~bias]), k
        assert np.abs(a[bias] - b[bias]).max() <= 1e-6 * np.abs(a[bias]).max(), k
    assert np.isfinite(res["persistent"]["hv"]).all() and np.abs(res["persistent"]["hv"]).max() > 0
    assert rel(res["persistent"]["hv2"], res["general"]["hv2"]) < 1e-6      # (a product of the slightly different g)


def test_layerwise_fvp_takes_a_direction_at_any_4_byte_offset():
    """on the layer-wise path (256 / 512-wide policies) a direction that is a view at an odd offset of a larger
    tensor is served by the generic output-layer chain instead of being refused by the one-pass output layer (which reads
    it with 16-byte loads): same product as with an aligned copy of the same values."""
    import torch
    from mjrl_amd.engine import UpdateEngine
    n, m, hid, N = 40, 5, (256, 256), 1500
    rng = np.random.RandomState(4)
    th = synth.perturbed_params(synth.init_params(n, m, hid), scale=0.05)
    ident = np.concatenate([np.zeros(n), np.ones(n), np.zeros(m), np.ones(m)]).astype(np.float32)
    eng = UpdateEngine(n, m, hid)
    assert not eng.fused
    eng.set_policy(th, th, ident, ident)
    eng.set_batch(rng.randn(N, n), rng.randn(N, m), rng.randn(N))
    eng.surr_vpg()
    big = torch.from_numpy(rng.randn(th.size + 8).astype(np.float32)).to(eng.device)
    view = big[1:1 + th.size]
    assert view.data_ptr() % 16 != 0
    out_view = eng.fvp(view).clone()
    out_copy = eng.fvp(view.clone()).clone()
    assert rel(out_view.cpu().numpy(), out_copy.cpu().numpy()) < 1e-6
    hvo = O.fvp(th.astype(np.float64), eng.obs.cpu().numpy().astype(np.float64), view.cpu().numpy().astype(np.float64), n, m, hid, O.Transforms(n, m))
    assert rel(out_view.cpu().numpy(), hvo) < 1e-5
    eng.close()


def test_empty_shard_contributes_zero():
    """a rank that holds no trajectories (N_local == 0, N_global > 0) must produce zeros on both paths"""
    import torch
    from mjrl_amd.engine import UpdateEngine
    for n, m, hid in ((17, 6, (64, 64)), (40, 9, (96, 48))):
        th = synth.perturbed_params(synth.init_params(n, m, hid))
        tr = np.concatenate([np.zeros(n), np.ones(n), np.zeros(m), np.ones(m)]).astype(np.float32)
        eng = UpdateEngine(n, m, hid)
        eng.set_policy(th, th, tr, tr)
        eng.set_batch(np.zeros((0, n), np.float32), np.zeros((0, m), np.float32), np.zeros(0, np.float32), N_global=1000)
        g, s = eng.surr_vpg()
        assert float(g.abs().max()) == 0.0 and s == 0.0
        assert float(eng.fvp(torch.ones(th.size, device=eng.device)).abs().max()) == 0.0
        assert eng.eval_surr_kl() == (0.0, 0.0)
        eng.close()


def test_eval_reads_k1_observation_image_only_under_the_same_input_transform():
    """K3 after K1 takes the normalised observations from the image K1 left in the forward-activation cache (no staging,
    no normalisation of the raw block) -- results are bit-identical to the raw path (MJX_K3_XIMG is read once per process,
    so the two paths are compared through the kernel's own guard): when the NEW policy's input transform is changed in
    place after K1, the kernel must notice (it compares with the snapshot) and normalise the raw observations itself."""
    import torch
    from mjrl_amd.engine import UpdateEngine
    n, m, hid, N = 17, 6, (64, 64), 6000 + 7
    rng = np.random.RandomState(21)
    obs, act, adv = rng.randn(N, n), rng.randn(N, m), rng.randn(N)
    th = synth.perturbed_params(synth.init_params(n, m, hid))
    step = (0.02 * rng.randn(th.size)).astype(np.float32)
    tr = O.Transforms(n, m, 0.1 * rng.randn(n), 1 + 0.1 * rng.rand(n), 0.05 * rng.randn(m), 1 + 0.2 * rng.rand(m))
    pk = np.concatenate([tr.in_shift, tr.in_scale, tr.out_shift, tr.out_scale]).astype(np.float32)
    tr2 = O.Transforms(n, m, tr.in_shift + 0.05, tr.in_scale * 1.1, tr.out_shift, tr.out_scale)
    pk2 = np.concatenate([tr2.in_shift, tr2.in_scale, tr2.out_shift, tr2.out_scale]).astype(np.float32)
    t_new, t_old = (th + step).astype(np.float64), th.astype(np.float64)
    eng = UpdateEngine(n, m, hid)
    assert eng.fused
    eng.set_policy(th, th, pk, pk)
    eng.set_batch(obs, act, adv)
    eng.surr_vpg()                                        # K1: activation cache incl. the observation image, old outputs, snapshot
    eng.theta_new.copy_(torch.from_numpy(th + step).to(eng.device))
    eng.old_is_new = False
    eng._bind_policy()
    s1, k1 = eng.eval_surr_kl()                           # image path
    assert abs(s1 - O.surrogate(t_new, t_old, obs, act, adv, n, m, hid, tr, tr)) < 2e-6
    k1t = O.mean_kl(t_new, t_old, obs, n, m, hid, tr, tr)
    assert abs(k1 - k1t) < 1e-5 * k1t
    eng.tr_new.copy_(torch.from_numpy(pk2).to(eng.device))     # in place: no bind call tells the library
    s2, k2 = eng.eval_surr_kl()                           # the image belongs to another transform: raw path
    assert abs(s2 - O.surrogate(t_new, t_old, obs, act, adv, n, m, hid, tr2, tr)) < 2e-6
    k2t = O.mean_kl(t_new, t_old, obs, n, m, hid, tr2, tr)
    assert abs(k2 - k2t) < 1e-5 * k2t and abs(k2 - k1) > 1e-3 * k1t
    eng.tr_new.copy_(torch.from_numpy(pk).to(eng.device))      # back: the image applies again, same bits as before
    assert eng.eval_surr_kl() == (s1, k1)
    eng.close()


def test_eval_reuses_old_policy_outputs_only_when_unchanged():
    """K3 after K1 takes the old policy's means / log-likelihoods from what K1 stored -- but only while the
    old parameters still equal the snapshot taken then; an in-place change of theta_old must be noticed by
    the kernel itself (it compares before trusting), and mjx_bind_batch drops the stored outputs."""
    import torch
    from mjrl_amd.engine import UpdateEngine
    n, m, hid, N = 17, 6, (64, 64), 4000 + 13
    rng = np.random.RandomState(11)
    obs, act, adv = rng.randn(N, n), rng.randn(N, m), rng.randn(N)
    th = synth.perturbed_params(synth.init_params(n, m, hid))
    step = (0.02 * rng.randn(th.size)).astype(np.float32)
    ident = np.concatenate([np.zeros(n), np.ones(n), np.zeros(m), np.ones(m)]).astype(np.float32)
    tr64 = O.Transforms(n, m)

    def truth(t_new, t_old):
        s = O.surrogate(t_new.astype(np.float64), t_old.astype(np.float64), obs, act, adv, n, m, hid, tr64, tr64)
        k = O.mean_kl(t_new.astype(np.float64), t_old.astype(np.float64), obs, n, m, hid, tr64, tr64)
        return s, k

    eng = UpdateEngine(n, m, hid)
    eng.set_policy(th, th, ident, ident)
    eng.set_batch(obs, act, adv)
    eng.surr_vpg()                                        # K1: stores the old-policy outputs + parameter snapshot
    eng.theta_new.copy_(torch.from_numpy(th + step).to(eng.device))
    eng.old_is_new = False
    eng._bind_policy()
    s1, k1 = eng.eval_surr_kl()                           # uses the stored outputs
    st, kt = truth(th + step, th)
    assert abs(s1 - st) < 2e-6 and abs(k1 - kt) < 1e-5 * kt
    # now move theta_old in place: the stored outputs are stale, the kernel has to recompute the old forward
    th_old2 = (th - step).astype(np.float32)
    eng.theta_old.copy_(torch.from_numpy(th_old2).to(eng.device))
    s2, k2 = eng.eval_surr_kl()
    st2, kt2 = truth(th + step, th_old2)
    assert abs(s2 - st2) < 2e-6 and abs(k2 - kt2) < 1e-5 * kt2
    assert abs(k2 - k1) > 1e-3 * kt                        # (the two situations really differ)
    # prefix re-binding keeps the stored outputs valid for the leading rows
    eng.theta_old.copy_(torch.from_numpy(th).to(eng.device))
    Np = 2500
    eng.bind_rows(Np, N_global=Np)
    s3, k3 = eng.eval_surr_kl()
    s3t = O.surrogate((th + step).astype(np.float64), th.astype(np.float64), obs[:Np], act[:Np], adv[:Np], n, m, hid, tr64, tr64)
    k3t = O.mean_kl((th + step).astype(np.float64), th.astype(np.float64), obs[:Np], n, m, hid, tr64, tr64)
    assert abs(s3 - s3t) < 2e-6 and abs(k3 - k3t) < 1e-5 * k3t
    # a new batch in the same engine: nothing stored applies any more
    obs2 = rng.randn(N, n)
    eng.set_batch(obs2, act, adv)
    s4, k4 = eng.eval_surr_kl()
    s4t = O.surrogate((th + step).astype(np.float64), th.astype(np.float64), obs2, act, adv, n, m, hid, tr64, tr64)
    assert abs(s4 - s4t) < 2e-6
    # ... and the Fisher-vector product after a new batch must not read the previous batch's activations
    eng.theta_new.copy_(torch.from_numpy(th).to(eng.device)); eng.old_is_new = True; eng._bind_policy()
    v = rng.randn(th.size).astype(np.float32)
    hv = eng.fvp(torch.from_numpy(v).to(eng.device)).cpu().numpy()
    hvt = O.fvp(th.astype(np.float64), obs2, v.astype(np.float64), n, m, hid, tr64)
    assert rel(hv, hvt) < TOL_FVP
    eng.close()


def test_path_stager_gpu_exact_and_update_unchanged():
    """Page-locked staging + chunked side-stream upload + device cast delivers exactly astype(float32) of the
    concatenated paths, and NPG.train_from_paths (which ingests through it) matches the reference fixture
    (covered by test_npg_agent_update_vs_reference); here: ragged paths, reuse of the stager across batches."""
    import torch
    from mjrl_amd.engine import UpdateEngine
    rng = np.random.RandomState(4)
    eng = UpdateEngine(17, 6, (64, 64))
    for rep, nt in enumerate((37, 120, 11)):
        lens = rng.randint(1, 1000, size=nt)
        paths = [dict(observations=rng.randn(T, 17), actions=rng.randn(T, 6)) for T in lens]
        out = eng.stage_paths(paths)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out["observations"].cpu().numpy(), np.concatenate([p["observations"] for p in paths]).astype(np.float32))
        np.testing.assert_array_equal(out["actions"].cpu().numpy(), np.concatenate([p["actions"] for p in paths]).astype(np.float32))
    eng.close()


@pytest.mark.parametrize("hid", [(64, 64), (32, 32), (128, 64, 32), ()])
def test_batched_policy_forward_device(hid):
    """policy.model.forward on a CUDA batch (learned-model rollouts / evaluation,
    model_accel/sampling.py:66-89) goes through mjx_policy_forward; == the NumPy network / fp64 oracle."""
    import torch
    from mjrl_amd.policies.gaussian_mlp import MLP
    from mjrl_amd.policies.gaussian_linear import LinearPolicy
    n, m, N = 11, 3, 5003

    class Spec:
        observation_dim, action_dim, horizon = n, m, 100
    pol = MLP(Spec, hidden_sizes=hid, seed=3, init_log_std=-0.5) if hid else LinearPolicy(Spec, seed=3)
    rng = np.random.RandomState(8)
    pol.set_param_values((pol.get_param_values() + 0.1 * rng.randn(pol.get_param_values().size)).astype(np.float32))
    ish, isc = rng.randn(n).astype(np.float32), (0.5 + rng.rand(n)).astype(np.float32)
    osh, osc = rng.randn(m).astype(np.float32), (0.5 + rng.rand(m)).astype(np.float32)
    pol.model.set_transformations(ish, isc, osh, osc)
    obs = rng.randn(N, n).astype(np.float32)
    dev = pol.model.forward(torch.from_numpy(obs).cuda())
    assert dev.is_cuda and dev.shape == (N, m)
    host = pol.model.forward(obs)                                   # NumPy fp32 network (get_action's path)
    truth = O.forward(pol.get_param_values().astype(np.float64), obs.astype(np.float64), n, m, tuple(hid),
                      O.Transforms(n, m, ish, isc, osh, osc))
    assert np.abs(dev.cpu().numpy() - truth).max() < 2e-5 * max(1.0, np.abs(truth).max())
    assert np.abs(dev.cpu().numpy() - host).max() < 2e-5 * max(1.0, np.abs(truth).max())
    # the old network follows its own parameters / transforms
    old = pol.old_model.forward(torch.from_numpy(obs[:100]).cuda()).cpu().numpy()
    assert np.abs(old - pol.old_model.forward(obs[:100])).max() < 2e-5 * max(1.0, np.abs(truth).max())
    import pickle
    pickle.loads(pickle.dumps(pol))                                 # the device context never travels


# N3 bars (r04: set from what the chains measure, VERDICT r03 weak 3 -- they had been 2e-3 of the parameter movement "because minibatch
# Adam is chaotic"; measured on MI355X: BC-MSE 6.4e-7, BC-MLE 5.7e-7 (54 / 20 steps), PPO 3.2e-6 / 2.5e-6 (28 steps, twice), PPO against
# torch autograd + torch.optim.Adam 9.7e-6, the persistent one-workgroup trainer against the per-step launches 7e-6..1e-5).  The bars
# sit ~10 x above that; the long chains (MLP baseline: 188 / 9 372 steps) stay statistical, as documented there.
BAR_BC = {"bc_mse_32x32": 1e-5, "bc_mle_64x64": 1e-5}
BAR_PPO = 3e-5


@pytest.mark.parametrize("name", ["bc_mse_32x32", "bc_mle_64x64"])
def test_bc_minibatch_adam_vs_reference(name):
    """BC.train: the device minibatch-Adam loop lands on the parameters the reference's torch loop
    reaches from the same start with the same np.random.choice stream (behavior_cloning.py:107-136)."""
    from mjrl_amd.algos.behavior_cloning import BC
    from mjrl_amd.policies.gaussian_mlp import MLP
    g = load(name)
    n, m, hid = int(g["n"]), int(g["m"]), tuple(int(h) for h in g["hidden"])

    class Spec:
        observation_dim, action_dim, horizon = n, m, int(g["T"])
    pol = MLP(Spec, hidden_sizes=hid, seed=1, init_log_std=-0.5)
    pol.set_param_values(g["theta0"])
    paths = synth.make_paths(int(g["n_traj"]), int(g["T"]), n, m, seed=int(g["seed_paths"]))
    bc = BC(paths, pol, epochs=int(g["epochs"]), batch_size=int(g["mb"]), lr=float(g["lr"]), loss_type=str(g["loss_type"]),
            save_logs=True, set_transforms=bool(int(g["set_transforms"])))
    assert rel(pol.get_param_values(), g["theta_start"]) < 1e-6
    np.testing.assert_allclose(pol.model.in_scale, g["in_scale"], rtol=1e-6)
    np.random.seed(int(g["seed_np"]))
    bc.train()
    moved = np.linalg.norm(g["theta_final"] - g["theta_start"])
    err = np.linalg.norm(pol.get_param_values() - g["theta_final"])
    print("[N3 %s] distance from the reference's parameters / their movement: %.2e" % (name, err / moved))
    assert err < BAR_BC[name] * moved, (err, moved)
    assert np.array_equal(pol.get_param_values(), pol.get_old_param_values())
    assert bc.logger.log['loss_after'][-1] < bc.logger.log['loss_before'][-1]


def test_bc_with_a_caller_supplied_adam():
    """behavior_cloning.py:42: BC(optimizer=torch.optim.Adam(policy.trainable_params, lr=...)) -- the caller's optimizer drives the
    device loop when it is the Adam the loop implements: its learning rate is used, its moments / step count are taken over and
    written back after every fit.  Two fits with the caller's optimizer == two fits of a BC built with lr=..., bit for bit, and the
    optimizer's state afterwards IS the device loop's state (a torch step from it continues the same chain)."""
    import torch
    from mjrl_amd.algos.behavior_cloning import BC
    from mjrl_amd.policies.gaussian_mlp import MLP
    spec = type("Spec", (), dict(observation_dim=11, action_dim=3, horizon=40))
    paths = synth.make_paths(12, 40, 11, 3, seed=3)
    out = {}
    for mode in ("own", "callers"):
        pol = MLP(spec, hidden_sizes=(32, 32), seed=1, init_log_std=-0.5)
        opt = torch.optim.Adam(pol.trainable_params, lr=7e-4) if mode == "callers" else None
        bc = BC(paths, pol, epochs=2, batch_size=64, lr=7e-4, optimizer=opt, loss_type='MLE', save_logs=False)
        np.random.seed(5)
        bc.train(); bc.train()
        out[mode] = (pol.get_param_values().copy(), bc, opt, pol)
    assert np.array_equal(out["own"][0], out["callers"][0])
    bc, opt, pol = out["callers"][1:]
    steps = 2 * 2 * (12 * 40 // 64)
    st = opt.state[pol.trainable_params[0]]
    assert int(float(st["step"])) == steps and st["exp_avg"].shape == pol.trainable_params[0].shape and float(st["exp_avg_sq"].abs().sum()) > 0
    flat_m = torch.cat([opt.state[p]["exp_avg"].reshape(-1) for p in pol.trainable_params]).numpy()
    assert np.array_equal(flat_m, bc._adam[0].cpu().numpy())
    with pytest.raises(NotImplementedError):
        BC(paths, pol, optimizer=torch.optim.SGD(pol.trainable_params, lr=1e-3), save_logs=False)
    with pytest.raises(NotImplementedError):
        BC(paths, pol, optimizer=torch.optim.Adam(pol.trainable_params, lr=1e-3, weight_decay=1e-2), save_logs=False)


def test_ppo_minibatch_adam_vs_reference():
    """PPO.train_from_paths twice (the Adam state carries over) == the reference (ppo_clip.py:59-110)."""
    from mjrl_amd.algos.ppo_clip import PPO
    from mjrl_amd.policies.gaussian_mlp import MLP
    g = load("ppo_64x64")
    n, m, hid = int(g["n"]), int(g["m"]), tuple(int(h) for h in g["hidden"])

    class Spec:
        observation_dim, action_dim, horizon = n, m, int(g["T"])
    pol = MLP(Spec, hidden_sizes=hid, seed=1, init_log_std=-0.5)
    pol.set_param_values(g["theta0"])
    paths = synth.make_paths(int(g["n_traj"]), int(g["T"]), n, m, seed=int(g["seed_paths"]))
    rng = np.random.RandomState(int(g["seed_adv"]))
    for p in paths:
        p["advantages"] = rng.randn(len(p["rewards"])) * 2.0 + 0.3
    agent = PPO(None, pol, None, clip_coef=float(g["clip"]), epochs=int(g["epochs"]), mb_size=int(g["mb"]), learn_rate=float(g["lr"]))
    np.random.seed(int(g["seed_np"]))
    for ref in (g["theta_after_1"], g["theta_after_2"]):
        agent.train_from_paths(paths)
        moved = np.linalg.norm(ref - g["theta0"])
        err = np.linalg.norm(pol.get_param_values() - ref)
        print("[N3 ppo_64x64] distance from the reference's parameters / their movement: %.2e" % (err / moved))
        assert err < BAR_PPO * moved, (err, moved)
    assert agent.last_update["kl_dist"] > 0


def test_ppo_fixed_old_policy_vs_torch_autograd():
    """PPO with the old policy held fixed during the epochs (reference_aliasing=False, the algorithm as published):
    the device loop == torch autograd + torch.optim.Adam driven through the policy's differentiable CPU mirror with the
    same minibatches (the gradient of min(LR adv, clamp(LR) adv) incl. the clipped branches, the log_std block, Adam)."""
    import torch
    from mjrl_amd.algos.ppo_clip import PPO
    from mjrl_amd.policies.gaussian_mlp import MLP
    g = load("ppo_64x64")
    n, m, hid = int(g["n"]), int(g["m"]), tuple(int(h) for h in g["hidden"])

    class Spec:
        observation_dim, action_dim, horizon = n, m, int(g["T"])
    paths = synth.make_paths(int(g["n_traj"]), int(g["T"]), n, m, seed=int(g["seed_paths"]))
    rng = np.random.RandomState(int(g["seed_adv"]))
    for p in paths:
        p["advantages"] = rng.randn(len(p["rewards"])) * 2.0 + 0.3
    obs = np.concatenate([p["observations"] for p in paths]); act = np.concatenate([p["actions"] for p in paths])
    adv = np.concatenate([p["advantages"] for p in paths]); adv = (adv - adv.mean()) / (adv.std() + 1e-6)
    mb, epochs, lr, clip = int(g["mb"]), int(g["epochs"]), float(g["lr"]), float(g["clip"])
    # torch replica
    ref = MLP(Spec, hidden_sizes=hid, seed=1, init_log_std=-0.5); ref.set_param_values(g["theta0"])
    opt = torch.optim.Adam(ref.trainable_params, lr=lr)
    np.random.seed(123)
    for _ in range(epochs * (obs.shape[0] // mb)):
        idx = np.random.choice(obs.shape[0], size=mb)
        ad = torch.from_numpy(np.float32(adv[idx]))
        LR = ref.likelihood_ratio(ref.new_dist_info(obs[idx], act[idx]), ref.old_dist_info(obs[idx], act[idx]))
        surr = torch.mean(torch.min(LR * ad, torch.clamp(LR, 1 - clip, 1 + clip) * ad))
        opt.zero_grad(); (-surr).backward(); opt.step()
    # device
    pol = MLP(Spec, hidden_sizes=hid, seed=1, init_log_std=-0.5); pol.set_param_values(g["theta0"])
    agent = PPO(None, pol, None, clip_coef=clip, epochs=epochs, mb_size=mb, learn_rate=lr, reference_aliasing=False)
    np.random.seed(123)
    agent.train_from_paths(paths)
    moved = np.linalg.norm(ref.get_param_values() - g["theta0"])
    err = np.linalg.norm(pol.get_param_values() - ref.get_param_values())
    print("[N3 ppo vs torch autograd, old policy fixed] %.2e of the movement" % (err / moved))
    assert moved > 0.05 and err < 1e-4 * moved, (err, moved)


@pytest.mark.parametrize("n,m,H,B", [(17, 6, 64, 64), (5, 2, 32, 8), (63, 16, 64, 32), (11, 3, 32, 64), (17, 6, 64, 48), (39, 16, 64, 12),
                                     (33, 7, 64, 20)])
def test_persistent_policy_trainer_equals_launch_path(n, m, H, B, monkeypatch):
    """mjx_policy_minibatch_adam has two implementations: one persistent workgroup for small nets / minibatches
    (csrc/policy_fit.h) and ~19 launches per step for everything else.  Same inputs -> same parameters, Adam moments and
    per-step losses (to fp32 summation-order noise), for MSE, MLE and both flavours of the clipped surrogate, with
    non-trivial input / output transforms and ragged feature counts."""
    import torch
    from mjrl_amd.engine import UpdateEngine
    from mjrl_amd._lib import check, ptr
    rng = np.random.RandomState(n * 100 + B)
    hid, N, steps = (H, H), 5000, 24
    eng = UpdateEngine(n, m, hid)
    th0 = synth.perturbed_params(synth.init_params(n, m, hid))
    tho0 = (th0 + 0.02 * rng.randn(th0.size)).astype(np.float32)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    obs, act, adv = dev(rng.randn(N, n).astype(np.float32)), dev(rng.randn(N, m).astype(np.float32)), dev(rng.randn(N).astype(np.float32))
    idx = dev(rng.randint(0, N, size=(steps, B)).astype(np.int32))
    mk_tr = lambda: dev(np.concatenate([0.1 * rng.randn(n), 1 + 0.2 * rng.rand(n), 0.1 * rng.randn(m), 1 + 0.2 * rng.rand(m)]).astype(np.float32))
    tr, tro = mk_tr(), mk_tr()
    for loss, track in ((0, 1), (1, 1), (2, 1), (2, 0)):
        out = []
        for no_fit in ("0", "1"):
            monkeypatch.setenv("MJX_NO_POLICY_FIT", no_fit)
            th, tho = dev(th0.copy()), dev(tho0.copy())
            am, av = torch.zeros_like(th), torch.zeros_like(th)
            lt = torch.zeros(steps, dtype=torch.float64).cuda()
            for part in range(2):                                  # two calls: the Adam state carries over (step0)
                check(eng.lib.mjx_policy_minibatch_adam(eng.ctx, loss, ptr(obs), ptr(act), ptr(adv), ptr(idx[part * 12:]), 12, B, ptr(th),
                                                        ptr(tr), ptr(tho), ptr(tro), track, ptr(am), ptr(av), part * 12, 3e-4, 0.2,
                                                        ptr(lt[part * 12:]), eng.stream()))
            torch.cuda.synchronize()
            out.append([t.cpu().numpy().astype(np.float64) for t in (th, am, av, lt)])
        (th_a, am_a, av_a, lt_a), (th_b, am_b, av_b, lt_b) = out
        moved = np.linalg.norm(th_b - th0)
        assert moved > 1e-3
        print("[N3 persistent vs launches, loss %d track %d] %.2e of the movement" % (loss, track, np.linalg.norm(th_a - th_b) / moved))
        assert np.linalg.norm(th_a - th_b) < 1e-4 * moved, (loss, track, np.linalg.norm(th_a - th_b), moved)
        np.testing.assert_allclose(lt_a, lt_b, rtol=2e-4, atol=1e-5)      # (a PPO minibatch loss is a sum of cancelling O(1) terms)
        assert np.linalg.norm(am_a - am_b) < 1e-3 * np.linalg.norm(am_b)
        assert np.linalg.norm(av_a - av_b) < 1e-3 * np.linalg.norm(av_b)


def test_update_is_bit_reproducible_and_handles_odd_sizes():
    """Fixed-order reductions everywhere: the same update twice gives identical bits (no atomics, no dependence on
    dispatch order); batch sizes that are not tile multiples (1, 31, 33, 4097 samples) agree with the oracle."""
    import torch
    from mjrl_amd.engine import UpdateEngine
    n, m, hid = 17, 6, (64, 64)
    rng = np.random.RandomState(21)
    th = synth.perturbed_params(synth.init_params(n, m, hid))
    ident = np.concatenate([np.zeros(n), np.ones(n), np.zeros(m), np.ones(m)]).astype(np.float32)
    tr64 = O.Transforms(n, m)
    eng = UpdateEngine(n, m, hid)
    eng.set_policy(th, th, ident, ident)
    N = 200003
    obs, act, adv = rng.randn(N, n).astype(np.float32), rng.randn(N, m).astype(np.float32), rng.randn(N).astype(np.float32)
    outs = []
    for rep in range(2):
        eng.set_batch(obs, act, adv)
        g, s = eng.surr_vpg()
        x, gx = eng.cg_solve(g, 10, 1e-4)
        outs.append((g.cpu().numpy().copy(), x.cpu().numpy().copy(), s, gx))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert outs[0][2] == outs[1][2] and outs[0][3] == outs[1][3]
    for Ns in (1, 31, 33, 4097):
        o, a, ad = obs[:Ns], act[:Ns], adv[:Ns]
        eng.set_batch(o, a, ad)
        g, s = eng.surr_vpg()
        truth = O.vpg(th.astype(np.float64), th.astype(np.float64), o.astype(np.float64), a.astype(np.float64), ad.astype(np.float64), n, m, hid, tr64, None)
        assert rel(g.cpu().numpy(), truth) < TOL_VPG, Ns
        v = torch.from_numpy(truth.astype(np.float32)).to(eng.device)
        hv = eng.fvp(v).cpu().numpy()
        hvt = O.fvp(th.astype(np.float64), o.astype(np.float64), truth, n, m, hid, tr64)
        assert rel(hv, hvt) < TOL_FVP, Ns
    eng.close()


def test_one_upload_per_batch_shared_by_update_and_baselines():
    """baseline.predict / the policy update / baseline.fit of one iteration read ONE device copy of the observations
    (utils/ingest.stage_shared); a new batch gets new tensors and leaves the previous ones intact."""
    import torch
    from mjrl_amd.baselines.quadratic_baseline import QuadraticBaseline
    from mjrl_amd.engine import UpdateEngine
    from mjrl_amd.utils import ingest
    ingest.drop_shared()
    n, m = 17, 6
    spec = type("Spec", (), dict(observation_dim=n, action_dim=m, horizon=100))
    rng = np.random.RandomState(2)

    def mk():
        return [dict(observations=rng.randn(T, n), actions=rng.randn(T, m), rewards=rng.randn(T), returns=rng.randn(T)) for T in (50, 77, 100)]
    paths = mk()
    bl = QuadraticBaseline(spec)
    bl.fit(paths)
    pred = bl.predict(paths[1])
    eng = UpdateEngine(n, m, (64, 64))
    st = eng.stage_paths(paths)
    reg = ingest._SHARED[("cuda", torch.cuda.current_device())]
    assert st["observations"].data_ptr() == reg["observations"]["f32"].data_ptr()
    up1 = reg["observations"]["raw"].data_ptr()
    bl.fit(paths)                                           # same batch again: no new upload
    assert reg["observations"]["raw"].data_ptr() == up1
    np.testing.assert_array_equal(st["observations"].cpu().numpy(), np.concatenate([p["observations"] for p in paths]).astype(np.float32))
    keep = st["observations"]
    snapshot = keep.cpu().numpy().copy()
    paths2 = mk()
    st2 = eng.stage_paths(paths2)
    assert st2["observations"].data_ptr() != keep.data_ptr()
    np.testing.assert_array_equal(keep.cpu().numpy(), snapshot)          # the earlier batch is untouched while referenced
    np.testing.assert_array_equal(st2["observations"].cpu().numpy(), np.concatenate([p["observations"] for p in paths2]).astype(np.float32))
    assert np.all(np.isfinite(pred))
    eng.close()


def test_large_d_cg_update_on_many_workgroups_breaks_like_the_reference():
    """r06: beyond 8 192 parameters the CG vector update runs as three launches of 64 workgroups (csrc/vecops.h k_cgm_*); the
    residual test of cg_solve.py:19-20 (`break`) is decided on the device: with a tolerance the solve reaches after a few
    iterations every later product must leave x untouched -- same x as the oracle's loop, which really breaks."""
    import torch
    from mjrl_amd.engine import UpdateEngine
    n, m, hid, N = 23, 5, (128, 128), 6000 + 7
    rng = np.random.RandomState(77)
    obs, act, adv = rng.randn(N, n), rng.randn(N, m), rng.randn(N)
    th = synth.perturbed_params(synth.init_params(n, m, hid), scale=0.05)
    assert th.size > 8192
    ident = np.concatenate([np.zeros(n), np.ones(n), np.zeros(m), np.ones(m)]).astype(np.float32)
    eng = UpdateEngine(n, m, hid)
    assert not eng.fused
    eng.set_policy(th, th, ident, ident)
    eng.set_batch(obs, act, adv)
    g, _ = eng.surr_vpg()
    g64 = g.cpu().numpy().astype(np.float64)
    hv = lambda p: O.fvp(th.astype(np.float64), obs, p, n, m, hid, O.Transforms(n, m), damping=0.1)
    # the residual after each of 12 iterations (fp64): pick a tolerance that the k-th iteration is the first to reach
    res, x, r = [], np.zeros_like(g64), g64.copy()
    p_, rr = r.copy(), r.dot(r)
    for _ in range(12):
        z = hv(p_); a = rr / p_.dot(z); x += a * p_; r -= a * z
        new = r.dot(r); p_ = r + (new / rr) * p_; rr = new; res.append(rr)
    k = next(i for i in range(3, 10) if res[i] < 0.5 * min(res[:i]))    # (the residual is not monotone: a drop below everything before it)
    tol = float(np.sqrt(res[k] * min(res[:k])))
    assert res[k] < tol < min(res[:k])
    x_ref = O.cg_solve(hv, g64, 12, residual_tol=tol)
    x_dev, _ = eng.cg_solve(g, 12, 0.1, tol=tol)
    assert rel(x_dev.cpu().numpy(), x_ref) < TOL_STEP
    x_all = O.cg_solve(hv, g64, 12)                               # (the unbroken solve really differs)
    assert rel(x_all, x_ref) > 10 * TOL_STEP
    eng.close()


def test_device_step_length_matches_host_formula():
    """mjx_apply_npg_step: alpha = sqrt(|delta / (g.x + 1e-20)|) formed on the device (fp64) and applied == the host
    formula of npg_cg.py:133 followed by mjx_apply_step, bit for bit, incl. the log_std clamp."""
    import torch
    from mjrl_amd.engine import UpdateEngine
    n, m, hid = 17, 6, (64, 64)
    rng = np.random.RandomState(3)
    th = synth.perturbed_params(synth.init_params(n, m, hid))
    th[-m:] = -2.9999                                            # a step may push log_std below the clamp
    ident = np.concatenate([np.zeros(n), np.ones(n), np.zeros(m), np.ones(m)]).astype(np.float32)
    eng = UpdateEngine(n, m, hid)
    eng.set_policy(th, th, ident, ident)
    eng.set_batch(rng.randn(5000, n).astype(np.float32), rng.randn(5000, m).astype(np.float32), rng.randn(5000).astype(np.float32))
    g, _ = eng.surr_vpg()
    x, gx = eng.cg_solve(g, 10, 1e-4)
    eng.apply_npg_step(0.05, -3.0)
    dev = eng.theta_new.cpu().numpy().copy()
    late = eng.deferred()
    alpha = np.sqrt(np.abs(0.05 / (gx + 1e-20)))
    assert late["gdotx"] == gx and late["alpha"] == alpha
    eng.apply_step(alpha, -3.0)
    assert np.array_equal(dev, eng.theta_new.cpu().numpy())
    assert dev[-m:].min() >= -3.0
    eng.close()
