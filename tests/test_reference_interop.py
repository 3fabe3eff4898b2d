"""Drop-in checks against the UNMODIFIED reference (only where /root/reference exists, i.e. in the build
container): the reference's own agents drive this package's Policy object and get bit-identical results,
and its torch-optimizer algorithms (BC, PPO: SURVEY N3) train it."""
import copy
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/mjrl"), reason="reference tree not present")


@pytest.fixture(scope="module")
def ref():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import _ref_import
    _ref_import.install()
    import torch
    torch.set_num_threads(4)
    from mjrl.utils.gym_env import EnvSpec
    return EnvSpec


def _paths(n, m):
    from oracle import synth
    paths = synth.make_paths(10, 50, n, m, seed=0)
    rng = np.random.RandomState(5)
    for p in paths:
        p["advantages"] = rng.randn(len(p["rewards"]))
    return paths


def test_reference_npg_and_trpo_drive_our_policy_bit_identically(ref):
    from mjrl.algos.npg_cg import NPG as RefNPG
    from mjrl.algos.trpo import TRPO as RefTRPO
    from mjrl.policies.gaussian_mlp import MLP as RefMLP
    from mjrl_amd.policies.gaussian_mlp import MLP
    spec = ref(11, 3, 100)
    for cls, kw in ((RefNPG, dict(normalized_step_size=0.05)), (RefTRPO, dict(kl_dist=0.01))):
        ours = MLP(spec, hidden_sizes=(32, 32), seed=3, init_log_std=-0.5)
        theirs = RefMLP(spec, hidden_sizes=(32, 32), seed=3, init_log_std=-0.5)
        cls(None, ours, None, **kw).train_from_paths(_paths(11, 3))
        cls(None, theirs, None, **kw).train_from_paths(_paths(11, 3))
        assert np.array_equal(ours.get_param_values(), theirs.get_param_values())
        assert ours.old_equals_new()


def test_reference_bc_and_ppo_train_our_policy(ref):
    from mjrl.algos.behavior_cloning import BC
    from mjrl.algos.ppo_clip import PPO
    from mjrl_amd.policies.gaussian_mlp import MLP
    spec = ref(11, 3, 100)
    pol = MLP(spec, hidden_sizes=(32, 32), seed=3, init_log_std=-0.5)
    paths = _paths(11, 3)
    for p in paths:
        p["actions"] = np.tanh(p["observations"][:, :3])
    np.random.seed(1)
    for loss in ("MSE", "MLE"):
        bc = BC(paths, pol, epochs=3, batch_size=32, lr=1e-3, loss_type=loss, set_transforms=(loss == "MSE"))
        bc.train(suppress_fit_tqdm=True)
        assert bc.logger.log['loss_after'][-1] < bc.logger.log['loss_before'][-1]
        assert pol.old_equals_new()
    assert not np.allclose(pol.model.in_scale, 1.0)            # BC's set_transforms reached both nets
    np.testing.assert_array_equal(pol.model.in_scale, pol.old_model.in_scale)
    before = pol.get_param_values()
    ppo = PPO(None, pol, None, epochs=1, mb_size=64, learn_rate=3e-4)
    ppo.train_from_paths(_paths(11, 3))
    assert not np.array_equal(before, pol.get_param_values()) and pol.old_equals_new()
    # the optimiser's in-place steps landed in the NumPy store get_action reads
    o = np.random.RandomState(0).randn(11)
    np.random.seed(2); a = pol.get_action(o)[1]['mean']
    import torch
    b = pol.model(torch.from_numpy(np.float32(o.reshape(1, -1)))).detach().numpy().ravel()
    np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6)


def test_our_baselines_have_the_reference_surface(ref):
    import inspect
    from mjrl.baselines.mlp_baseline import MLPBaseline as R1
    from mjrl.baselines.quadratic_baseline import QuadraticBaseline as R2
    from mjrl.baselines.linear_baseline import LinearBaseline as R3
    from mjrl.algos.npg_cg import NPG as R4
    from mjrl.algos.trpo import TRPO as R5
    from mjrl.algos.dapg import DAPG as R6
    from mjrl.policies.gaussian_mlp import MLP as R7
    from mjrl_amd.baselines.mlp_baseline import MLPBaseline as O1
    from mjrl_amd.baselines.quadratic_baseline import QuadraticBaseline as O2, LinearBaseline as O3
    from mjrl_amd.algos.npg_cg import NPG as O4
    from mjrl_amd.algos.trpo import TRPO as O5
    from mjrl_amd.algos.dapg import DAPG as O6
    from mjrl_amd.policies.gaussian_mlp import MLP as O7
    for r, o in ((R1, O1), (R2, O2), (R3, O3), (R4, O4), (R5, O5), (R6, O6), (R7, O7)):
        rp, op = inspect.signature(r.__init__).parameters, inspect.signature(o.__init__).parameters
        assert list(rp) == list(op), (r, list(rp), list(op))                       # same names, same order
        for k in rp:
            if rp[k].default is not inspect._empty:
                assert rp[k].default == op[k].default, (r, k)
    for name in ("train_step", "train_from_paths", "process_paths", "CPI_surrogate", "kl_old_new", "flat_vpg", "HVP", "build_Hvp_eval"):
        assert hasattr(O4, name)
    for name in ("get_action", "get_param_values", "set_param_values", "mean_LL", "log_likelihood", "old_dist_info",
                 "new_dist_info", "likelihood_ratio", "mean_kl"):
        assert hasattr(O7, name)
