"""VERDICT r03 item 7, part (ii): what would split-bf16 products do to the numbers?

The cached Fisher-vector-product kernel sits at 95 % of its serial-issue bound in native fp32; the only > 1.3x lever left on 83 %
of an update is the 16x faster bf16 MFMA pipe with every fp32 operand split into bf16 pieces (hi / mid / lo) and the product
formed from the piece products that matter, accumulated in fp32.  Before any kernel is written: the ERROR of that arithmetic,
emulated exactly on the CPU (bf16 x bf16 is exact in fp32, the accumulation is fp32 like the MFMA's), for one Fisher-vector
product and one 10-iteration CG solve at the BASELINE configs[1] shapes, against fp64 truth, next to native fp32's.

    python tools/probe_split_error.py [N=100000]

modes: f32 (native), bf16x3 (3 pieces, the 6 products with i + j <= 2), bf16x3_all9, bf16x2 (2 pieces, 3 products -- what a
"cheaper" split would give: lower precision than the reference, listed for contrast only).
(Nothing here imports oracle/: tools are measurement infrastructure, the FVP is restated inline.)
"""
import json
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from tools._synth import init_params, perturbed_params  # noqa: E402

n, m, H = 17, 6, (64, 64)


def bf16_round(x):
    """fp32 -> nearest bf16 (round to nearest even), returned as fp32"""
    b = x.astype(np.float32).view(np.uint32)
    b = (b + np.uint32(0x7FFF) + ((b >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return b.view(np.float32)


def split(x, pieces):
    out, r = [], x.astype(np.float32)
    for _ in range(pieces):
        p = bf16_round(r)
        out.append(p)
        r = r - p                               # exact in fp32
    return out


def mm(a, b, mode):
    """a @ b with the arithmetic of `mode`; operands fp32 (fp64 for mode f64)"""
    if mode == "f64":
        return a @ b
    a, b = a.astype(np.float32), b.astype(np.float32)
    if mode == "f32":
        return a @ b
    pieces = 2 if mode == "bf16x2" else 3
    A, B = split(a, pieces), split(b, pieces)
    lim = {"bf16x3": 2, "bf16x3_all9": 4, "bf16x2": 1}[mode]
    terms = [(i + j, A[i] @ B[j]) for i in range(pieces) for j in range(pieces) if i + j <= lim]
    acc = np.zeros((a.shape[0], b.shape[1]), np.float32)
    for _, t in sorted(terms, key=lambda q: -q[0]):       # small terms first, one fp32 accumulator (as chained MFMAs would)
        acc = acc + t
    return acc


def unflatten(th):
    sizes = (n,) + H + (m,)
    Ws, bs, o = [], [], 0
    for i in range(3):
        k = sizes[i + 1] * sizes[i]
        Ws.append(th[o:o + k].reshape(sizes[i + 1], sizes[i])); o += k
        bs.append(th[o:o + sizes[i + 1]]); o += sizes[i + 1]
    return Ws, bs, th[o:]


def fvp(th, obs, v, mode, chunk=4096):
    """Gauss-Newton Fisher-vector product of mean_kl at theta_new == theta_old (mjrl/algos/npg_cg.py:62-81 restated: SURVEY 8a-a9);
    weight gradients accumulate per `chunk` samples in the mode's arithmetic and across chunks in fp64 (the kernel: per wave in MFMA
    accumulators, then fp64)"""
    dt = np.float64 if mode == "f64" else np.float32
    th, v, obs = th.astype(dt), v.astype(dt), obs.astype(dt)
    Ws, bs, s = unflatten(th)
    Vs, cs, vs = unflatten(v)
    N = obs.shape[0]
    u = np.exp(s) ** 2
    eps = dt(1e-8)
    D = (2.0 / (2.0 * u + eps)).astype(dt)
    gW = [np.zeros(W.shape, np.float64) for W in Ws]
    gb = [np.zeros(b.shape, np.float64) for b in bs]
    for lo in range(0, N, chunk):
        x = obs[lo:lo + chunk]
        h1 = np.tanh(mm(x, Ws[0].T, mode) + bs[0]).astype(dt)
        h2 = np.tanh(mm(h1, Ws[1].T, mode) + bs[1]).astype(dt)
        t1 = ((mm(x, Vs[0].T, mode) + cs[0]) * (1 - h1 * h1)).astype(dt)
        t2 = ((mm(h1, Vs[1].T, mode) + mm(t1, Ws[1].T, mode) + cs[1]) * (1 - h2 * h2)).astype(dt)
        mud = (mm(h2, Vs[2].T, mode) + mm(t2, Ws[2].T, mode) + cs[2]).astype(dt)
        d3 = (D * mud / dt(N)).astype(dt)
        d2 = (mm(d3, Ws[2], mode) * (1 - h2 * h2)).astype(dt)
        d1 = (mm(d2, Ws[1], mode) * (1 - h1 * h1)).astype(dt)
        for l, (d_, a_) in enumerate(((d1, x), (d2, h1), (d3, h2))):
            gW[l] += mm(d_.T, a_, mode).astype(np.float64)
            gb[l] += d_.sum(axis=0, dtype=np.float64)
    c = 16.0 * u * u / (2.0 * u + eps) ** 2 - 4.0 * u / (2.0 * u + eps)
    return np.concatenate([np.concatenate([gW[l].ravel(), gb[l]]) for l in range(3)] + [(c * vs).astype(np.float64)]).astype(dt)


def cg(A, b, iters=10):
    """mjrl/utils/cg_solve.py:3-22 with fp64 dot products (the device loop's arithmetic)"""
    x, r, p = np.zeros_like(b), b.copy(), b.copy()
    rr = float(r.astype(np.float64) @ r.astype(np.float64))
    for _ in range(iters):
        z = A(p)
        a = b.dtype.type(rr / float(p.astype(np.float64) @ z.astype(np.float64)))
        x = x + a * p
        r = r - a * z
        nrr = float(r.astype(np.float64) @ r.astype(np.float64))
        p = r + b.dtype.type(nrr / rr) * p
        rr = nrr
    return x


def rel(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    rng = np.random.RandomState(0)
    th = perturbed_params(init_params(n, m, H))
    obs = rng.randn(N, n)
    g = rng.randn(th.size).astype(np.float32)
    g *= 1.0 / np.linalg.norm(g)
    damping = 1e-4
    out = {"N": N, "shapes": "obs 17, act 6, 64x64 (BASELINE configs[1])", "cg_iters": 10, "damping": damping}
    h64 = fvp(th, obs, g, "f64")
    x64 = cg(lambda p: fvp(th, obs, p, "f64") + damping * p, g.astype(np.float64))
    for mode in ("f32", "bf16x3", "bf16x3_all9", "bf16x2"):
        h = fvp(th, obs, g, mode)
        x = cg(lambda p: fvp(th, obs, p, mode) + np.float32(damping) * p, g.astype(np.float32))
        out[mode] = {"fvp_rel_l2_vs_f64": rel(h, h64), "cg10_rel_l2_vs_f64": rel(x, x64)}
        print(mode, out[mode], flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
