"""Rebuild the inputs of a golden fixture from its seeds (see tests/golden/make_golden.py)."""
import os

import numpy as np

from oracle import npg_oracle as O
from oracle import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def fake_advantages(paths, seed):
    rng = np.random.RandomState(seed)
    for p in paths:
        p["advantages"] = rng.randn(len(p["rewards"])) * 2.0 + 0.3


class NpgCase:
    def __init__(self, name):
        g = self.g = load(name)
        self.name = name
        self.n, self.m = int(g["n"]), int(g["m"])
        self.hidden = tuple(int(h) for h in g["hidden"])
        self.cg_iters = int(g["cg_iters"])
        self.big = bool(g.get("big", False))
        self.paths = synth.make_paths(int(g["n_traj"]), int(g["T"]), self.n, self.m, seed=int(g["path_seed"]),
                                      ragged=bool(g["ragged"]))
        fake_advantages(self.paths, int(g["adv_seed"]))
        self.obs = np.concatenate([p["observations"] for p in self.paths])
        self.act = np.concatenate([p["actions"] for p in self.paths])
        adv = np.concatenate([p["advantages"] for p in self.paths])
        self.adv_w = O.whiten(adv)
        self.wide = bool(g.get("wide", False))      # N >= d fixtures of make_golden_big.py (reference + fp64-oracle vectors)
        if self.big:
            th = synth.init_params(self.n, self.m, self.hidden, seed=1, init_log_std=-0.5)
            self.theta0 = synth.perturbed_params(th, scale=0.02)
        else:
            self.theta0 = g["theta0"].astype(np.float32)
        self.tr = None
        if "in_shift" in g:
            self.tr = (g["in_shift"], g["in_scale"], g["out_shift"], g["out_scale"])
        self.demo_paths = None
        if "demo_n_traj" in g:
            self.demo_paths = synth.make_paths(int(g["demo_n_traj"]), int(g["demo_T"]), self.n, self.m,
                                               seed=int(g["demo_seed"]))

    def transforms(self, dtype=np.float64):
        if self.tr is None:
            return None
        return O.Transforms(self.n, self.m, *self.tr, dtype=dtype)

    def check(self, key, v, rtol):
        """relative-L2 check of a d-vector against the fixture (strided when the fixture is 'big')."""
        g = self.g
        v = np.asarray(v, np.float64)
        if self.big:
            ref, mine = g[key + "_sub"].astype(np.float64), v[::int(g["stride"])]
            err = np.linalg.norm(mine - ref) / np.linalg.norm(ref)
            nerr = abs(np.linalg.norm(v) - float(g[key + "_norm"])) / float(g[key + "_norm"])
            assert nerr < rtol, (self.name, key, "norm", nerr)
        else:
            ref = g[key].astype(np.float64)
            err = np.linalg.norm(v - ref) / np.linalg.norm(ref)
        assert err < rtol, (self.name, key, err)
        return err


    def check_step(self, key, v, tol):
        """The north-star bar on a step-direction vector of a `wide` fixture: rel-L2 to the REFERENCE's fp32 result
        < tol, norm included -- no fallback (r03: the "distance to fp64 truth" escape of earlier rounds is gone; the fp64
        numbers are still returned for the record).  -> dict of the measured errors."""
        g = self.g
        s = int(g["stride"])
        v = np.asarray(v, np.float64)
        ref, f64 = g[key + "_sub"].astype(np.float64), g[key + "_f64_sub"].astype(np.float64)
        e_ref = float(np.linalg.norm(v[::s] - ref) / np.linalg.norm(ref))
        e_f64 = float(np.linalg.norm(v[::s] - f64) / np.linalg.norm(f64))
        ref_f64 = float(g["err_ref_vs_f64_" + key])
        n_ref = abs(np.linalg.norm(v) - float(g[key + "_norm"])) / float(g[key + "_norm"])
        out = dict(vs_reference=e_ref, vs_fp64=e_f64, reference_vs_fp64=ref_f64, norm_vs_reference=n_ref)
        assert e_ref < tol and n_ref < tol, (self.name, key, out)
        return out


NPG_CASES = ["npg_cfg1_linear", "npg_pointmass_32x32", "npg_cfg2_small", "npg_cfg2_ragged_tr"]
BIG_CASES = ["npg_cfg4_small"]
