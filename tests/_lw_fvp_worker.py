"""Worker of test_persistent_gemm_bitwise_equals_general_kernel: one layer-wise Fisher-vector product + gradient with
whatever MJX_LW_* switches the parent set in the environment; writes the result vectors."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from oracle import synth
    from mjrl_amd.engine import UpdateEngine
    out, n, m, h1, h2, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    rng = np.random.RandomState(3)
    hid = (h1, h2)
    th = synth.perturbed_params(synth.init_params(n, m, hid), scale=0.05)
    obs, act, adv = rng.randn(N, n).astype(np.float32), rng.randn(N, m).astype(np.float32), rng.randn(N).astype(np.float32)
    ident = np.concatenate([np.zeros(n), np.ones(n), np.zeros(m), np.ones(m)]).astype(np.float32)
    eng = UpdateEngine(n, m, hid)
    assert not eng.fused
    eng.set_policy(th, th, ident, ident)
    eng.set_batch(obs, act, adv)
    g = eng.surr_vpg()[0].cpu().numpy()
    v = rng.randn(th.size).astype(np.float32)
    hv = eng.fvp(torch.from_numpy(v).to(eng.device)).cpu().numpy()
    hv2 = eng.fvp(torch.from_numpy(g).to(eng.device)).cpu().numpy()
    np.savez(out, g=g, hv=hv, hv2=hv2)
    eng.close()


if __name__ == "__main__":
    main()
