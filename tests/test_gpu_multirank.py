"""SURVEY 8e beyond train_from_paths: a whole `train_step` on two ranks equals the one-rank run -- ONE baseline fit over all
ranks' trajectories (mjrl/algos/batch_reinforce.py:94-110, baselines/quadratic_baseline.py:44-69, mlp_baseline.py:61-95),
statistics over all paths -- and a lost rank surfaces as an error, not as NaN parameters.  Two PROCESSES share the one GPU of the
box (torch.distributed.run, gloo for set-up traffic; RCCL refuses two ranks per device): the update's rank sums run inside libmjx
over its peer exchange, exactly as they would between two GPUs."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def _run(worker, args, world, port, extra_env=None, timeout=280):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    script = os.path.join(ROOT, "tests", worker)
    if world == 1:
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        cmd = [sys.executable, script] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(port), script] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("kind", ["quadratic", "mlp"])
def test_two_rank_train_step_equals_one_rank(tmp_path, kind):
    """2 x train_step(N = 2 001 trajectories of 50 steps: shares of 1 000 / 1 001) with NPG + the quadratic / the MLP baseline.  Every rank samples
    its contiguous share with the seeds a single process uses for those episodes, so the ranks' paths in rank order ARE the
    one-process batch (NPG with 5 CG iterations: BASELINE configs[0]); after iteration 1 (identical inputs) the policy step matches
    to 1e-5 (measured 2e-7), the ridge baseline's predictions to 1e-9 (3e-15) and the MLP baseline's parameters bit for bit (all-gathered block in rank order, the last rank's permutation); the
    logged statistics -- VF errors, return statistics, sample count -- are those of the whole batch on every rank."""
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    port = 29700 + (os.getpid() % 200) + (50 if kind == "mlp" else 0)
    _run("_two_rank_train_step_worker.py", [one, kind], 1, port)
    _run("_two_rank_train_step_worker.py", [two, kind], 2, port)
    a, b = np.load(one), np.load(two)
    assert bool(b["ranks_identical"][0]), "ranks must hold bit-identical policies, baselines and logs"
    assert str(b["comm_kind"][0]) == "peer" and str(a["comm_kind"][0]) == "None"
    assert int(a["seed"][0]) == int(b["seed"][0]) == 2 + 2 * 2001
    np.testing.assert_array_equal(a["theta0"], b["theta0"])
    # iteration 1: the same trajectories on both sides
    assert rel(b["grad1"], a["grad1"]) < 2e-6, rel(b["grad1"], a["grad1"])            # a plain sum over samples and ranks
    s1 = rel(b["theta1"].astype(np.float64) - a["theta0"], a["theta1"].astype(np.float64) - a["theta0"])
    assert s1 < 1e-5, (s1, rel(b["x1"], a["x1"]))
    np.testing.assert_allclose(b["stats1"], a["stats1"], rtol=1e-12)                 # [mean, std, min, max] of ALL returns, N
    # log: alpha, kl, surr_improvement, running_score, num_samples, VF_error_before / after, stoc_pol_*
    np.testing.assert_allclose(b["log1"][[0, 1, 2]], a["log1"][[0, 1, 2]], rtol=2e-5)
    np.testing.assert_allclose(b["log1"][[3, 4, 7, 8, 9, 10]], a["log1"][[3, 4, 7, 8, 9, 10]], rtol=1e-12)
    assert b["log1"][4] == 2001 * 50
    if kind == "quadratic":
        assert rel(b["pred1"], a["pred1"]) < 1e-9, rel(b["pred1"], a["pred1"])
        np.testing.assert_allclose(b["log1"][[5, 6]], a["log1"][[5, 6]], rtol=1e-9)          # VF errors over all ranks' paths
        assert rel(b["bl1"], a["bl1"]) < 1e-6                                                # (coefficients: conditioning of the normal equations)
    else:
        np.testing.assert_array_equal(b["bl1"], a["bl1"])                                    # the identical trainer on the identical block
        np.testing.assert_array_equal(b["pred1"], a["pred1"])
        np.testing.assert_array_equal(b["log1"][[5, 6]], a["log1"][[5, 6]])
    # iteration 2 starts from policies 2e-7 apart (sampling included): still the same update (measured 3e-7)
    s2 = rel(b["theta2"].astype(np.float64) - b["theta1"], a["theta2"].astype(np.float64) - a["theta1"])
    assert s2 < 1e-5, s2
    np.testing.assert_allclose(b["stats2"], a["stats2"], rtol=1e-5)
    # (the ridge fit is a smooth function of its inputs: 2e-9; the ReLU / Adam chain of the MLP baseline is not -- 3 000 steps from
    #  inputs 1e-7 apart end 2e-3 apart in prediction, in any implementation: tools/probe_fit_wide.py)
    assert rel(b["pred2"], a["pred2"]) < (1e-6 if kind == "quadratic" else 2e-2)
    print("[two-rank train_step, %s baseline] step 1 %.2e, step 2 %.2e, prediction 1 %.2e, prediction 2 %.2e"
          % (kind, s1, s2, rel(b["pred1"], a["pred1"]) if kind == "quadratic" else 0.0, rel(b["pred2"], a["pred2"])))


def test_peer_timeout_surfaces_as_an_error(tmp_path):
    """a rank that stops sending: the bounded wait (MJX_PEER_TIMEOUT_MS) ends, the engine raises MjxError instead of handing NaN
    parameters to policy.set_param_values, and tears the transport down"""
    out = str(tmp_path / "timeout.npz")
    _run("_peer_timeout_worker.py", [out], 2, 29950 + (os.getpid() % 40), extra_env={"MJX_PEER_TIMEOUT_MS": "150", "MJX_PEER_COMM": "1"})
    r = np.load(out)
    assert str(r["comm_kind"][0]) == "peer" and np.all(np.isfinite(r["good"]))
    msg = str(r["raised"][0])
    assert "timed out" in msg and "peer exchange" in msg, msg
    assert str(r["comm_kind_after"][0]) == "None"
    assert float(r["seconds"][0]) < 20.0


def test_nccl_group_without_in_library_rccl_falls_back_to_the_hook(tmp_path):
    """engine._native_comm on an nccl group whose in-library communicator cannot be created (librccl not loadable by libmjx):
    the ranks agree, detach, and attach the hook transport -- the one-call update loops keep running, over torch.distributed's
    own RCCL group -- and a one-rank group reproduces the single-process update bit for bit"""
    out = str(tmp_path / "fallback.npz")
    _run("_rccl_fallback_worker.py", [out], 1, 0, extra_env={"MJX_RCCL_DISABLE": "1", "MJX_TEST_PORT": str(29800 + (os.getpid() % 100))})
    r = np.load(out)
    assert str(r["kind"][0]) == "hook", r["kind"]
    assert bool(r["same"][0])


@pytest.mark.parametrize("world,fault", [(2, "slot"), (2, "flag"), (3, "slot")])
def test_a_corrupted_peer_transport_is_rejected_before_the_first_update(tmp_path, world, fault):
    """engine._transport_self_test: at attach, one known-answer rank sum (d floats + 4 doubles, rank-dependent, exactly summable)
    through the transport, compared with the closed form bit for bit and across the ranks.  MJX_PEER_FAULT makes ONE rank's producer
    misbehave -- `slot`: its vectors land in another rank's slot at the peers (a mis-mapped buffer); `flag`: it never raises its
    arrival flags (a lost flag store; the consumers' bounded wait ends in NaN).  Every rank must reject the peer exchange, fall
    through to the hook, and the update must be the clean run's (which runs on the peer exchange)."""
    clean, bad = str(tmp_path / "clean.npz"), str(tmp_path / "bad.npz")
    port = 29400 + (os.getpid() % 150) + 7 * world + (3 if fault == "flag" else 0)
    _run("_peer_fault_worker.py", [clean], world, port)
    _run("_peer_fault_worker.py", [bad], world, port + 1,
         extra_env={"MJX_PEER_FAULT": fault, "MJX_PEER_FAULT_RANK": str(world - 1), "MJX_PEER_TIMEOUT_MS": "200"})
    a, b = np.load(clean), np.load(bad)
    assert str(a["comm_kind"][0]) == "peer" and not bool(a["warned"][0])
    assert str(b["comm_kind"][0]) == "hook" and bool(b["warned"][0]), (b["comm_kind"], b["warned"])
    assert bool(a["ranks_identical"][0]) and bool(b["ranks_identical"][0])
    assert np.all(np.isfinite(b["theta"]))
    assert rel(b["theta"], a["theta"]) < 1e-6, rel(b["theta"], a["theta"])          # the same update over the other transport


def test_two_rank_ppo_equals_one_rank(tmp_path):
    """PPO under a process group (r05; it refused before): the minibatch epochs are a sequential chain over rows of the WHOLE batch,
    so every rank gathers all ranks' rows and runs the identical chain from the same index draws -- two ranks holding 17 + 23
    trajectories end two iterations with the parameters of the one-process run (the advantage whitening is a rank sum: fp64
    statistics, the fp32 advantages agree to the last bit or one), bit-identical on both ranks."""
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    port = 29250 + (os.getpid() % 100)
    _run("_two_rank_ppo_worker.py", [one], 1, port)
    _run("_two_rank_ppo_worker.py", [two], 2, port)
    a, b = np.load(one), np.load(two)
    assert bool(b["ranks_identical"][0])
    th0 = None
    for k in ("theta1", "theta2"):
        d = rel(b[k], a[k])
        assert d < 1e-6, (k, d)
    np.testing.assert_allclose(b["stats"], a["stats"], rtol=1e-12)
    np.testing.assert_allclose(b["kl"], a["kl"], rtol=1e-4)
    np.testing.assert_allclose(b["surr"], a["surr"], rtol=1e-4, atol=1e-7)
