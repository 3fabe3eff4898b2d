"""CPU-side tests: the C-ABI library loads and exports every symbol include/mjx.h declares, host
classes behave like the reference's (pickle / deepcopy / RNG stream), and the product path fails
loudly without a GPU."""
import copy
import os
import pickle
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from mjrl_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "mjx.h")).read()
    declared = set(re.findall(r"\b(mjx_[a-z0-9_]+)\s*\(", hdr)) - {"mjx_allreduce_fn"}
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.mjx_version() >= 1
    assert lib.mjx_device_count() >= 0


def _parse_header_prototypes(hdr):
    """-> {name: (return type string, [parameter type strings])} for every function include/mjx.h declares"""
    src = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r"^\s*#.*$", " ", src, flags=re.M)
    out = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(mjx_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        ret, name, params = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        if ret.startswith("typedef"):
            continue
        plist = []
        if params not in ("", "void"):
            for p in params.split(","):
                p = p.strip()
                mm = re.match(r"(.*?)([A-Za-z_]\w*)?$", p)           # drop the parameter name
                t = mm.group(1).strip() if ("*" in p or " " in p) else p
                plist.append(" ".join(t.replace("*", " * ").split()))
        out[name] = (" ".join(ret.replace("*", " * ").split()), plist)
    return out


def test_ctypes_prototypes_match_the_header_argument_by_argument():
    """mjrl_amd/_lib.PROTOTYPES (the hand-written ctypes table) against a parse of include/mjx.h: the same functions, the
    same number of arguments, and per argument the same C scalar type (int / int64_t / float / double) or a pointer where
    the header has a pointer (ctx**, int*, char*, data pointers, function-pointer typedefs) -- so the table cannot drift
    from the ABI silently."""
    import ctypes
    from mjrl_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "mjx.h")).read()
    protos = _parse_header_prototypes(hdr)
    assert set(protos) == set(_lib.PROTOTYPES), set(protos) ^ set(_lib.PROTOTYPES)
    scalars = {"int": ctypes.c_int, "int64_t": ctypes.c_int64, "float": ctypes.c_float, "double": ctypes.c_double}
    fnptrs = {"mjx_allreduce_fn", "mjx_reduce_fn"}

    def is_pointer_ctype(t):
        return t in (ctypes.c_void_p, ctypes.c_char_p) or (isinstance(t, type) and issubclass(t, (ctypes._Pointer, ctypes._CFuncPtr)))

    for name, (ret, params) in protos.items():
        res, args = _lib.PROTOTYPES[name]
        assert len(args) == len(params), (name, params, args)
        if ret == "void":
            assert res is None, name
        elif "*" in ret:
            assert is_pointer_ctype(res), (name, ret, res)
        else:
            assert res is scalars[ret], (name, ret, res)
        for i, (p, a) in enumerate(zip(params, args)):
            base = p.replace("const ", "").strip()
            if "*" in base or base in fnptrs:
                assert is_pointer_ctype(a), (name, i, p, a)
                if base == "int *":
                    assert a in (ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)), (name, i, p, a)
                if base == "double *" and a is not ctypes.c_void_p:
                    assert a is ctypes.POINTER(ctypes.c_double), (name, i, p, a)
            else:
                assert a is scalars[base], (name, i, p, a)


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mjrl_amd import _lib
    from mjrl_amd.engine import UpdateEngine
    with pytest.raises(_lib.MjxError):
        UpdateEngine(17, 6, (64, 64))
    import ctypes
    lib = _lib.load()
    ctx = ctypes.c_void_p()
    rc = lib.mjx_create(ctypes.byref(ctx), 0, 17, 6, (ctypes.c_int * 2)(64, 64), 2)
    assert rc == -4 and b"not available" in lib.mjx_last_error()


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mjrl_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+(oracle|tests)\b", src, re.M), (dirpath, f)
                assert "import_module" not in src and "__import__" not in src, (dirpath, f)


def test_oracle_is_only_used_by_the_checkers():
    """outside tests/ only smoke() and bench.py's cpu_baseline leg may touch oracle/: the measurement tools do not, and
    in bench.py every oracle import sits inside cpu_baseline()"""
    pat = re.compile(r"^\s*(from|import)\s+oracle\b", re.M)
    for f in os.listdir(os.path.join(ROOT, "tools")):
        if f.endswith(".py"):
            assert not pat.search(open(os.path.join(ROOT, "tools", f)).read()), f
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def cpu_baseline("):src.index("def main(")]
    assert len(pat.findall(src)) == len(pat.findall(body)) >= 1


def _spec(n, m):
    return type("Spec", (), dict(observation_dim=n, action_dim=m, horizon=100))


def test_policy_surface_and_lifecycle():
    from mjrl_amd.policies.gaussian_mlp import MLP
    from mjrl_amd.policies.gaussian_linear import LinearPolicy
    p = MLP(_spec(17, 6), hidden_sizes=(64, 64), seed=7, init_log_std=-0.5)
    assert p.d == 17 * 64 + 64 + 64 * 64 + 64 + 64 * 6 + 6 + 6 == sum(p.param_sizes)
    th = p.get_param_values()
    assert th.dtype == np.float32 and th is not p.get_param_values()
    new = th.copy(); new[-6:] = -10.0
    p.set_param_values(new, set_new=True, set_old=False)
    assert np.all(p.get_param_values()[-6:] == -3.0) and not p.old_equals_new()      # clamp at min_log_std
    assert np.allclose(p.log_std_val, -3.0)
    p.set_param_values(new, set_new=True, set_old=True)
    assert p.old_equals_new()
    q, r = pickle.loads(pickle.dumps(p)), copy.deepcopy(p)
    o = np.random.RandomState(0).randn(17)
    np.random.seed(3); a = p.get_action(o)
    np.random.seed(3); b = q.get_action(o)
    np.random.seed(3); c = r.get_action(o)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[0], c[0])
    assert set(a[1]) == {"mean", "log_std", "evaluation"}
    lin = LinearPolicy(_spec(6, 2), seed=1)
    assert lin.d == 6 * 2 + 2 + 2 and lin.hidden_sizes == ()
    # host operators agree with the oracle
    from oracle import npg_oracle as O
    obs, act = np.random.RandomState(1).randn(50, 17), np.random.RandomState(2).randn(50, 6)
    ll = p.log_likelihood(obs, act)
    ll_o, _ = O.log_likelihood(p.get_param_values().astype(np.float64), obs, act, 17, 6, (64, 64))
    assert np.allclose(ll, ll_o, rtol=1e-5, atol=1e-5)
    assert abs(float(p.mean_kl(p.new_dist_info(obs, act), p.old_dist_info(obs, act)))) < 1e-6


def test_host_cg_matches_oracle():
    from mjrl_amd.utils.cg_solve import cg_solve
    from oracle import npg_oracle as O
    rng = np.random.RandomState(0)
    A = rng.randn(30, 30); A = A @ A.T + 0.1 * np.eye(30)
    b = rng.randn(30)
    x = cg_solve(lambda v: A @ v, b, x_0=b.copy(), cg_iters=7)
    np.testing.assert_allclose(x, O.cg_solve(lambda v: A @ v, b, 7), rtol=1e-12)


def test_agent_pickles_without_engine():
    from mjrl_amd.algos.npg_cg import NPG
    from mjrl_amd.algos.trpo import TRPO
    from mjrl_amd.algos.dapg import DAPG
    from mjrl_amd.policies.gaussian_mlp import MLP
    pol = MLP(_spec(5, 2), hidden_sizes=(32, 32), seed=0)
    for cls, kw in ((NPG, dict(normalized_step_size=0.05, hvp_sample_frac=0.5, input_normalization=0.9)),
                    (TRPO, dict(kl_dist=0.01)), (DAPG, dict(demo_paths=None, lam_0=0.1))):
        a = cls(None, pol, None, save_logs=True, seed=5, some_unknown_kwarg=1, **kw)
        b = pickle.loads(pickle.dumps(a))
        assert b.seed == 5 and b._engine_obj is None
    assert NPG(None, pol, None, kl_dist=0.02).n_step_size == 0.04          # kl_dist overrides (npg_cg.py:49)
    assert NPG(None, pol, None, input_normalization=1.5).input_normalization is None


def test_path_stager_matches_concatenate_cpu():
    """utils/ingest.PathStager (CPU fallback: plain memory, synchronous sends) == np.concatenate + astype(float32),
    for ragged fp64 paths, fp32 paths, the incremental interface and more than one group per call."""
    import torch
    from mjrl_amd.utils.ingest import PathStager

    class Backend:
        device = torch.device("cpu")
    Backend.torch = torch
    rng = np.random.RandomState(3)
    lens = [1, 7, 300, 64, 1000, 33]
    paths = [dict(observations=rng.randn(T, 5), actions=rng.randn(T, 2)) for T in lens]
    from mjrl_amd import _lib
    for threads, native in ((1, False), (4, False), (3, True)):
        be = Backend()
        be.lib = _lib.load() if native else None            # native: the gather runs in libmjx (mjx_host_gather)
        st = PathStager(be, threads=threads, group_rows=256)
        assert st.native == native
        out = st.stage(paths)
        assert out["observations"].dtype == torch.float32 and out["observations"].shape == (sum(lens), 5)
        np.testing.assert_array_equal(out["observations"].numpy(), np.concatenate([p["observations"] for p in paths]).astype(np.float32))
        np.testing.assert_array_equal(out["actions"].numpy(), np.concatenate([p["actions"] for p in paths]).astype(np.float32))
        # incremental: two add_paths calls, fp32 sources
        p32 = [dict(observations=p["observations"].astype(np.float32), actions=p["actions"].astype(np.float32)) for p in paths]
        st.begin(("observations", "actions"), (5, 2), (np.float32, np.float32), sum(lens))
        st.add_paths(p32[:2])
        st.add_paths(p32[2:])
        out = st.finish()
        np.testing.assert_array_equal(out["observations"].numpy(), np.concatenate([p["observations"] for p in p32]))
        # host-cast mode: the gather converts fp64 -> fp32 itself (mjx_host_gather_f64_f32): same bits as astype, no raw block
        if native:
            big = [dict(observations=rng.randn(T, 17) * 10.0 ** rng.randint(-30, 30, size=(T, 17)), actions=rng.randn(T, 6)) for T in (50000, 3, 70000)]
            big[0]["observations"][0, :4] = [np.inf, -np.inf, 1e-46, 3.4028235677973366e38]     # overflow / denormal / rounds-to-inf edge values
            out = st.stage(big, hostcast=True)
            np.testing.assert_array_equal(out["observations"].numpy(), np.concatenate([p["observations"] for p in big]).astype(np.float32))
            os.environ["MJX_NO_AVX2"] = "1"                          # the portable conversion loop gives the same bits
            try:
                out = st.stage(big, hostcast=True)
                np.testing.assert_array_equal(out["observations"].numpy(), np.concatenate([p["observations"] for p in big]).astype(np.float32))
            finally:
                del os.environ["MJX_NO_AVX2"]
            np.testing.assert_array_equal(out["actions"].numpy(), np.concatenate([p["actions"] for p in big]).astype(np.float32))
            assert st.raw("observations") is None and st._slots["observations"]["pin"].dtype == torch.float32
            out = st.stage(big, hostcast=("actions",))                # per key: observations keep their fp64 block
            assert st.raw("observations").dtype == torch.float64 and st.raw("actions") is None
            np.testing.assert_array_equal(st.raw("observations").numpy(), np.concatenate([p["observations"] for p in big]))
            np.testing.assert_array_equal(out["observations"].numpy(), np.concatenate([p["observations"] for p in big]).astype(np.float32))
        # capacity is enforced
        st.begin(("observations", "actions"), (5, 2), (np.float64, np.float64), 10)
        with pytest.raises(ValueError):
            st.add_paths(paths)
            st.finish()
        st.close()


def test_in_place_edits_between_uses_are_never_silently_ignored():
    """outside BatchREINFORCE.train_step an in-place edit of a per-path array between two uses of a batch
    must not be missed because it falls between the probed positions: user-owned arrays (rewards, observations) are
    compared exactly with the staged host copy, device-computed blocks (returns, advantages) come back as read-only
    views.  Inside train_step (ingest.trusted_iteration) the cheap identity + probe check applies."""
    import torch
    from mjrl_amd.utils import ingest
    ingest.drop_shared()
    h = ingest.DeviceHandle(torch, torch.device("cpu"), None)
    rng = np.random.RandomState(1)
    paths = [dict(observations=rng.randn(40, 3), rewards=rng.randn(40)) for _ in range(50)]     # > 32 paths: probes are sampled
    a = ingest.stage_shared(h, paths, ("rewards",))["rewards"]
    # path 1 is not among the probed paths (0, 2, 4, ...), element 5 not among the probed positions
    assert 1 not in list(ingest._probed(len(paths)))
    paths[1]["rewards"][5] += 3.0
    with ingest.trusted_iteration():
        assert ingest.lookup(h, paths, "rewards") is a["raw"]                    # (the cheap rule does not see it)
    assert ingest.lookup(h, paths, "rewards") is None                              # the exact rule does
    b = ingest.stage_shared(h, paths, ("rewards",))["rewards"]
    assert b["raw"][45].item() == paths[1]["rewards"][5]
    assert ingest.lookup(h, paths, "rewards") is b["raw"]
    # device-computed blocks: per-path views of one read-back, read-only
    from mjrl_amd.utils import process_samples as ps
    blk = torch.arange(2000, dtype=torch.float64)
    off = np.arange(0, 2001, 40)
    views = ps._hand_out(h, paths, "returns", blk, off)
    ingest.publish(h, paths, "returns", blk, views)
    assert ingest.lookup(h, paths, "returns") is blk
    with pytest.raises(ValueError):
        paths[3]["returns"][7] = 0.0
    with pytest.raises(ValueError):
        np.clip(paths[3]["returns"], -1, 1, out=paths[3]["returns"])
    paths[3]["returns"] = np.clip(paths[3]["returns"], -1, 1)                      # the way to edit: a new array -> the block is stale
    assert ingest.lookup(h, paths, "returns") is None
    ingest.drop_shared()


def test_staged_batch_registry_identity_rules():
    """utils/ingest: a batch is recognised by the IDENTITY of the path list and of every per-path array, held by
    strong references (ids of freed objects are recycled), plus probe values against in-place edits; device-computed
    blocks are published / looked up under the same rules (the id()-based fingerprint could hit falsely)."""
    import torch
    from mjrl_amd.utils import ingest
    ingest.drop_shared()
    h = ingest.DeviceHandle(torch, torch.device("cpu"), None)
    rng = np.random.RandomState(0)
    paths = [dict(observations=rng.randn(T, 5), rewards=rng.randn(T)) for T in (7, 3, 11)]
    a = ingest.stage_shared(h, paths, ("observations",))["observations"]
    b = ingest.stage_shared(h, paths, ("observations",))["observations"]
    assert a["f32"] is b["f32"] and a["raw"] is b["raw"]                       # same batch: one upload
    np.testing.assert_array_equal(a["raw"].numpy(), np.concatenate([p["observations"] for p in paths]))
    # a NEW list holding the same arrays is another batch (the temporary lists of baseline.predict(path))
    c = ingest.stage_shared(h, list(paths), ("observations",))["observations"]
    assert c["raw"] is not a["raw"]
    a = ingest.stage_shared(h, paths, ("observations",))["observations"]
    # one array replaced by an equal-valued copy: not the same batch
    paths[1]["observations"] = paths[1]["observations"].copy()
    d = ingest.stage_shared(h, paths, ("observations",))["observations"]
    assert d["raw"] is not a["raw"]
    # in-place edit of an array that is still the same object: caught by the probes
    paths[0]["observations"][0, 0] += 1.0
    e = ingest.stage_shared(h, paths, ("observations",))["observations"]
    assert e["raw"] is not d["raw"] and e["raw"][0, 0].item() == paths[0]["observations"][0, 0]
    # published device blocks
    blk = torch.arange(21, dtype=torch.float64)
    views = [blk.numpy()[0:7], blk.numpy()[7:10], blk.numpy()[10:21]]
    for p, v in zip(paths, views):
        p["returns"] = v
    assert ingest.lookup(h, paths, "returns") is None
    ingest.publish(h, paths, "returns", blk, views)
    assert ingest.lookup(h, paths, "returns") is blk
    calls = []
    assert ingest.derived(h, paths, "returns", "x", lambda: calls.append(1) or 42) == 42
    assert ingest.derived(h, paths, "returns", "x", lambda: calls.append(1) or 43) == 42 and len(calls) == 1
    paths[2]["returns"] = paths[2]["returns"].copy()
    assert ingest.lookup(h, paths, "returns") is None
    ingest.drop_shared_batch()
    assert ingest.lookup(h, paths, "observations") is None
    f = ingest.stage_shared(h, paths, ("observations",))["observations"]       # after the drop everything is staged afresh
    np.testing.assert_array_equal(f["raw"].numpy(), np.concatenate([p["observations"] for p in paths]))
    ingest.drop_shared()


def test_trusted_scope_reaches_the_helper_threads():
    """train_step's trusted scope is thread-local; the staging helper of _process_and_bind and the prefetch thread work on the
    caller's behalf and must inherit it (left untrusted they re-verified 184 MB element by element per iteration: 15 ms at 1M)."""
    import threading
    from mjrl_amd.utils import ingest
    seen = {}

    def job(trust):
        with trust():
            seen["helper"] = ingest._trusted()
    with ingest.trusted_iteration():
        t = threading.Thread(target=job, args=(ingest.carried_trust(),)); t.start(); t.join()
    assert seen["helper"] is True
    t = threading.Thread(target=job, args=(ingest.carried_trust(),)); t.start(); t.join()
    assert seen["helper"] is False


def test_raw_blocks_are_uploaded_on_demand():
    """utils/ingest.stage_shared(raw=...): a key whose raw (fp64) block nobody asked for goes up as fp32 only (converted by the
    native gather); the first raw request stages it again in full and is remembered for later batches; the in-place-edit
    rule holds for fp32-staged blocks too."""
    import torch
    from mjrl_amd import _lib
    from mjrl_amd.utils import ingest
    ingest.drop_shared()
    h = ingest.DeviceHandle(torch, torch.device("cpu"), _lib.load())
    rng = np.random.RandomState(2)
    paths = [dict(observations=rng.randn(T, 5), actions=rng.randn(T, 2)) for T in (40, 3, 11)]
    cat = lambda k: np.concatenate([p[k] for p in paths])
    a = ingest.stage_shared(h, paths, ("observations", "actions"), raw=())
    assert a["observations"]["raw"] is None and a["actions"]["raw"] is None
    np.testing.assert_array_equal(a["observations"]["f32"].numpy(), cat("observations").astype(np.float32))
    assert ingest.host_block(h, paths, "observations") is None                     # no host copy in the paths' dtype
    b = ingest.stage_shared(h, paths, ("observations",), raw=())["observations"]
    assert b["f32"] is a["observations"]["f32"]                                   # same batch: one upload
    c = ingest.stage_shared(h, paths, ("observations",))["observations"]          # a consumer of the raw block: staged again, in full
    np.testing.assert_array_equal(c["raw"].numpy(), cat("observations"))
    np.testing.assert_array_equal(c["f32"].numpy(), cat("observations").astype(np.float32))
    d = ingest.stage_shared(h, paths, ("observations",), raw=())["observations"]
    assert d["raw"] is c["raw"] and d["f32"] is c["f32"]                          # ... and serves the fp32 consumers from then on
    paths2 = [dict(observations=rng.randn(T, 5), actions=rng.randn(T, 2)) for T in (5, 6)]
    e = ingest.stage_shared(h, paths2, ("observations", "actions"), raw=())
    assert e["observations"]["raw"] is not None and e["actions"]["raw"] is None   # remembered per key
    # an in-place edit of an fp32-staged array is caught by the exact rule
    paths2[1]["actions"][2, 1] += 0.5
    f = ingest.stage_shared(h, paths2, ("actions",), raw=())["actions"]
    assert f["f32"] is not e["actions"]["f32"]
    assert f["f32"][5 + 2, 1].item() == np.float32(paths2[1]["actions"][2, 1])
    ingest.drop_shared()


def test_path_walk_extension_and_native_path_sums():
    """csrc/pathwalk.c (addresses / lengths of paths[.][key] by one C walk; identity of the staged arrays) and
    mjx_host_segment_sums (per-path reward sums on libmjx's host threads, added left to right: the bits of the reference's
    `sum(p["rewards"])`, mjrl/algos/batch_reinforce.py:187) -- host plumbing, runs without a GPU."""
    import ctypes
    from mjrl_amd import _lib
    from mjrl_amd import _pathwalk as pw
    from mjrl_amd.utils import ingest
    rng = np.random.RandomState(3)
    paths = [dict(observations=rng.randn(int(T), 5), rewards=rng.randn(int(T)) * 10 ** rng.uniform(-3, 3)) for T in rng.randint(1, 400, 300)]
    got = ingest.collect_arrays(paths, "observations")
    assert got is not None and got[2] == 5 and got[3] == 8
    assert all(int(got[0][i]) == p["observations"].ctypes.data and int(got[1][i]) == len(p["observations"]) for i, p in enumerate(paths))
    arrays = [p["rewards"] for p in paths]
    assert pw.identity(paths, "rewards", arrays) == 1
    # the rules of the fast walk: read-only arrays are fine, anything non-uniform / non-contiguous / not float sends the caller to its general route
    paths[3]["rewards"].setflags(write=False)
    assert ingest.collect_arrays(paths, "rewards") is not None
    for bad in (paths[7]["rewards"][::2], paths[7]["rewards"].astype(np.float32), paths[7]["rewards"].astype(np.int64), [1.0, 2.0]):
        q = list(paths); q[7] = dict(paths[7], rewards=bad)
        assert ingest.collect_arrays(q, "rewards") is None
        assert pw.identity(q, "rewards", arrays) == 0
    assert ingest.collect_arrays(paths, "no_such_key") is None and ingest.collect_arrays([], "rewards") is None
    # sums: every path left to right == Python's sum(), bit for bit (np.sum's pairwise order differs in the last bits)
    lib = _lib.load()
    ptrs, lens, w, isz = ingest.collect_arrays(paths, "rewards")
    for threads in (1, 7):
        out = np.empty(len(paths), np.float64)
        _lib.check(lib.mjx_host_segment_sums(ctypes.c_void_p(ptrs.ctypes.data), ctypes.c_void_p(lens.ctypes.data), len(paths),
                                             ctypes.c_void_p(out.ctypes.data), threads))
        ref = np.array([sum(p["rewards"]) for p in paths], np.float64)
        np.testing.assert_array_equal(out, ref)
    assert any(float(np.sum(p["rewards"])) != sum(p["rewards"]) for p in paths)      # (the two orders do differ on this data)


def test_bc_refuses_optimizers_the_device_loop_does_not_implement():
    """behavior_cloning.py:42 `optimizer=`: torch.optim.Adam over policy.trainable_params with Adam's default betas / eps and no weight
    decay is adopted (its lr); anything else is refused at construction -- there is no CPU training loop to fall back on"""
    import torch
    from mjrl_amd.algos.behavior_cloning import BC
    from mjrl_amd.policies.gaussian_mlp import MLP
    spec = type("S", (), dict(observation_dim=5, action_dim=2, horizon=5))
    pol = MLP(spec, hidden_sizes=(8, 8), seed=1)
    assert BC([], pol, optimizer=torch.optim.Adam(pol.trainable_params, lr=3e-4), save_logs=False).lr == pytest.approx(3e-4)
    for bad in (torch.optim.SGD(pol.trainable_params, lr=1e-3), torch.optim.Adam(pol.trainable_params, lr=1e-3, betas=(0.8, 0.9)),
                torch.optim.Adam([torch.nn.Parameter(torch.zeros(3))]), torch.optim.AdamW(pol.trainable_params)):
        with pytest.raises(NotImplementedError):
            BC([], pol, optimizer=bad, save_logs=False)


def test_native_permutation_is_numpys_stream_bit_for_bit():
    """mjx_host_mt19937_permutation == np.random.permutation(n) of the legacy global stream (utils/optimize_model.py:22 draws one per
    epoch): the same values AND the same generator state afterwards (a cached Gaussian included), for sizes around the mask / refill
    boundaries"""
    from mjrl_amd import _lib
    from mjrl_amd.baselines.mlp_baseline import _permutation_into
    lib = _lib.load()
    for seed in (0, 1, 123, 2 ** 31 - 1):
        for n in (0, 1, 2, 3, 5, 64, 623, 624, 625, 1000, 65536, 65537, 300007):
            np.random.seed(seed); np.random.randn(3)
            a = [np.random.permutation(n), np.random.permutation(n)]
            tail_a = (np.random.randn(2), np.random.randint(0, 1000, 5))
            np.random.seed(seed); np.random.randn(3)
            b = [np.empty(n, np.int32), np.empty(n, np.int32)]
            for o in b:
                _permutation_into(lib, o)
            tail_b = (np.random.randn(2), np.random.randint(0, 1000, 5))
            assert all(np.array_equal(x, y) for x, y in zip(a, b)), (seed, n)
            assert all(np.array_equal(x, y) for x, y in zip(tail_a, tail_b)), (seed, n)
    # r06: all epochs of a fit in ONE call (mjx_host_mt19937_permutations: generator and swaps on two threads above 65 536 rows) --
    # the same permutations, the same generator state afterwards
    from mjrl_amd.baselines.mlp_baseline import _permutations_into
    for seed, n, epochs in ((3, 1000, 3), (4, 65536, 2), (5, 65537, 3), (6, 300007, 2), (7, 1000003, 2), (8, 70001, 1)):
        np.random.seed(seed); np.random.randn(1)
        a = np.concatenate([np.random.permutation(n) for _ in range(epochs)])
        tail_a = np.random.randint(0, 10 ** 6, 4)
        np.random.seed(seed); np.random.randn(1)
        b = np.empty(n * epochs, np.int32)
        _permutations_into(lib, b, n, epochs)
        assert np.array_equal(a, b) and np.array_equal(tail_a, np.random.randint(0, 10 ** 6, 4)), (seed, n, epochs)


def test_native_minibatch_indices_are_numpys_choice_stream():
    """utils/ingest.minibatch_indices == np.stack([np.random.choice(N, size=mb) for _ in range(steps)]) (behavior_cloning.py:113,
    ppo_clip.py:77): same values, same generator state afterwards"""
    from mjrl_amd import _lib
    from mjrl_amd.utils.ingest import minibatch_indices
    lib = _lib.load()
    for seed in (0, 7, 99):
        for n, steps, mb in ((1, 3, 4), (2, 5, 3), (7, 11, 5), (1000, 40, 64), (65536, 9, 64), (65537, 9, 64), (1000003, 20, 64)):
            np.random.seed(seed); np.random.randn(1)
            a = np.stack([np.random.choice(n, size=mb) for _ in range(steps)]).astype(np.int32)
            ta = np.random.permutation(5)
            np.random.seed(seed); np.random.randn(1)
            b = minibatch_indices(lib, n, steps, mb)
            tb = np.random.permutation(5)
            assert np.array_equal(a, b) and np.array_equal(ta, tb), (seed, n, steps, mb)


def test_pending_log_entries_become_numbers(tmp_path):
    """utils/logger.py: a value logged before it exists (the errors / duration of a baseline fit still running on a side stream) --
    float() waits for it, get_current_log hands it on, save_log settles everything so log.csv / log.pickle hold numbers only, a
    pickle of the entry is a float"""
    from mjrl_amd.utils.logger import DataLog, PendingValue

    class Source:
        def __init__(self):
            self.hooks, self.is_done, self.waited = [], False, 0

        def finished(self):
            return self.is_done

        def result(self):
            self.waited += 1
            for h in self.hooks:
                h()
    log, src = DataLog(), Source()
    pv = PendingValue(src)
    src.hooks.append(lambda: pv.deliver(0.25))
    log.log_kv("alpha", 1.5); log.log_kv("VF_error_after", pv)
    assert str(pv) == "(fitting)" and src.waited == 0                    # printing does not wait
    row = log.get_current_log()
    assert row["alpha"] == 1.5 and isinstance(row["VF_error_after"], PendingValue) and src.waited == 0
    assert float(row["VF_error_after"]) == 0.25 and src.waited == 1      # reading the number does
    assert pv * 2 == 0.5 and 1 - pv == 0.75 and pickle.loads(pickle.dumps(pv)) == 0.25
    pv2 = PendingValue(src)
    src.hooks[:] = [lambda: pv2.deliver(0.5)]
    log.log_kv("alpha", 1.25); log.log_kv("VF_error_after", pv2)
    log.save_log(str(tmp_path))                                          # settles the one still pending
    assert log.log["VF_error_after"] == [pv, 0.5] or log.log["VF_error_after"] == [0.25, 0.5]
    rows = open(tmp_path / "log.csv").read().strip().splitlines()
    assert len(rows) == 3 and "fitting" not in "".join(rows) and rows[2].split(",")[-1] in ("0.5", "1.25")
    again = DataLog(); again.read_log(str(tmp_path / "log.csv"))
    assert [float(x) for x in again.log["VF_error_after"]] == [0.25, 0.5]
    src.is_done = True
    pv3 = PendingValue(src); src.hooks[:] = [lambda: pv3.deliver(2.0)]
    assert str(pv3) == "2.0"                                             # finished meanwhile: printing takes it over without waiting


def test_dropin_binds_the_reference_module_names(tmp_path):
    """mjrl_amd.dropin.install(): `from mjrl.algos.npg_cg import NPG` (examples/policy_opt_job_script.py:8-15) then yields this package's
    classes while mjrl.utils / mjrl.samplers stay the reference's -- in a process of its own (the aliases are process-wide)"""
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from oracle import ref_loader\n"
        "if ref_loader.install() is None: print('SKIP'); raise SystemExit\n"
        "from mjrl_amd import dropin; n = len(dropin.install())\n"
        "from mjrl.algos.npg_cg import NPG; from mjrl.policies.gaussian_mlp import MLP; from mjrl.baselines.mlp_baseline import MLPBaseline\n"
        "import mjrl.algos.trpo as t; from mjrl.utils.train_agent import train_agent; from mjrl.samplers.core import sample_paths\n"
        "print(n, NPG.__module__, MLP.__module__, MLPBaseline.__module__, t.TRPO.__module__, train_agent.__module__, sample_paths.__module__)\n"
    ) % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    if "SKIP" in r.stdout:
        pytest.skip("reference not available")
    assert r.stdout.split() == ["13", "mjrl_amd.algos.npg_cg", "mjrl_amd.policies.gaussian_mlp", "mjrl_amd.baselines.mlp_baseline",
                                "mjrl_amd.algos.trpo", "mjrl.utils.train_agent", "mjrl.samplers.core"], r.stdout


def test_path_walk_hand_out_equals_the_python_loop():
    """_pathwalk.hand_out: paths[i][key] = host[off[i]:off[i + 1]] in one C loop -- the same read-only views of the same block, the
    buffer's reference count following the views like it does for Python-made slices (ingest.download_owned recycles a page-locked
    block when its last view has died)"""
    import sys
    from mjrl_amd.utils import ingest
    if ingest._pathwalk is None or not hasattr(ingest._pathwalk, "hand_out"):
        pytest.skip("_pathwalk not built")
    rng = np.random.RandomState(0)
    lens = rng.randint(0, 50, 40)
    off = np.zeros(41, np.int64); np.cumsum(lens, out=off[1:])
    root = np.arange(int(off[-1]) + 8, dtype=np.float64)
    host = root[:int(off[-1])]
    host.setflags(write=False)
    paths = [dict(rewards=np.zeros(l)) for l in lens]
    before = sys.getrefcount(root)
    views = ingest._pathwalk.hand_out(paths, "returns", host, off)
    assert len(views) == 40 and all(paths[i]["returns"] is views[i] for i in range(40))
    for i in range(40):
        assert np.array_equal(views[i], host[off[i]:off[i + 1]]) and not views[i].flags.writeable and views[i].base is root
    assert sys.getrefcount(root) == before + 40
    del views
    for p in paths:
        p.pop("returns")
    assert sys.getrefcount(root) == before
    with pytest.raises((TypeError, ValueError)):
        ingest._pathwalk.hand_out([1, 2], "returns", host, off)


def test_allocator_tuning_is_opt_in_by_the_training_entry_points():
    """utils/ingest.tune_malloc: glibc's mmap / trim thresholds raised in the TRAINING process (what made releasing a rollout batch
    cost 0.5 or 13 ms) -- by BatchREINFORCE.train_step / dropin.install / bench.py, NOT by importing the package;
    idempotent; MJX_MALLOC_TUNE=0 leaves the allocator alone"""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import mjrl_amd; from mjrl_amd.utils import ingest; import mjrl_amd.algos.npg_cg, mjrl_amd.policies.gaussian_mlp;"
            "a = ingest.MALLOC_TUNED; b = ingest.tune_malloc(); c = ingest.tune_malloc(); print(a, b, c, ingest.MALLOC_TUNED)" % ROOT)
    env = {k: v for k, v in os.environ.items() if k != "MJX_MALLOC_TUNE"}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "None True True True", (r.stdout, r.stderr[-500:])
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(env, MJX_MALLOC_TUNE="0"), timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "None False False False", (r.stdout, r.stderr[-500:])
    import inspect
    from mjrl_amd import dropin
    from mjrl_amd.algos import batch_reinforce
    assert "tune_malloc()" in inspect.getsource(batch_reinforce.BatchREINFORCE.train_step)
    assert "tune_malloc()" in inspect.getsource(dropin.install)
