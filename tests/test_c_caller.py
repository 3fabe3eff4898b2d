"""A non-Python caller of the C ABI: tests/c/c_caller.c is compiled with gcc against include/mjx.h + libmjx.so only (no torch,
device memory through mjx_malloc / mjx_memcpy_*), runs one NPG update of the npg_cfg2_small fixture and must land on the
reference's update like the Python path does."""
import os
import struct
import subprocess

import numpy as np
import pytest

from tests._cases import NpgCase

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL_STEP = 1e-5


def build_c_caller(out_dir):
    import __graft_entry__ as ge
    ge.build()
    exe = os.path.join(str(out_dir), "c_caller")
    libdir = os.path.join(ROOT, "mjrl_amd", "csrc")
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "c_caller.c"),
           "-L", libdir, "-lmjx", "-lm", "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe]
    subprocess.check_call(cmd)
    return exe


def test_c_caller_compiles_and_links_against_the_header(tmp_path):
    """(CPU) the header is plain C99, every entry point the C program uses resolves against libmjx.so."""
    exe = build_c_caller(tmp_path)
    assert os.path.exists(exe)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "usage" in r.stderr


@pytest.mark.gpu
def test_c_caller_npg_update_vs_reference(tmp_path):
    c = NpgCase("npg_cfg2_small")
    exe = build_c_caller(tmp_path)
    N, d = c.obs.shape[0], c.theta0.size
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("8i", c.n, c.m, 2, c.hidden[0], c.hidden[1], N, d, c.cg_iters))
        f.write(struct.pack("2f", 1e-4, float(c.g["step"])))
        f.write(c.theta0.astype(np.float32).tobytes())
        f.write(c.obs.astype(np.float32).tobytes())
        f.write(c.act.astype(np.float32).tobytes())
        f.write(c.adv_w.astype(np.float32).tobytes())
    env = dict(os.environ)
    env.pop("PYTHONPATH", None)
    r = subprocess.run([exe, fin, fout], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    raw = open(fout, "rb").read()
    theta = np.frombuffer(raw[:4 * d], np.float32)
    res = np.frombuffer(raw[4 * d:], np.float64)
    step = theta.astype(np.float64) - c.theta0
    ref = c.g["new_params"].astype(np.float64) - c.theta0
    assert np.linalg.norm(step - ref) / np.linalg.norm(ref) < TOL_STEP
    assert abs(res[9] - float(c.g["alpha"])) < 1e-5 * float(c.g["alpha"])
    assert abs(res[1] / N - float(c.g["kl"])) < 1e-4 * float(c.g["kl"])
    # the same update through the Python host: bit-identical parameters (same library, same kernels, same order)
    from mjrl_amd.engine import UpdateEngine
    eng = UpdateEngine(c.n, c.m, c.hidden)
    ident = np.concatenate([np.zeros(c.n), np.ones(c.n), np.zeros(c.m), np.ones(c.m)]).astype(np.float32)
    eng.set_policy(c.theta0, c.theta0, ident, ident)
    eng.set_batch(c.obs, c.act, c.adv_w)
    eng.npg_update(c.cg_iters, 1e-4, float(c.g["step"]), -3.0)
    assert np.array_equal(eng.theta_new.cpu().numpy(), theta)
    eng.close()
