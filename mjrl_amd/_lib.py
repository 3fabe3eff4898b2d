"""ctypes binding of libmjx.so (C ABI: include/mjx.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).  There is
no CPU fallback: if the shared object is missing or no GPU is visible the product path
raises -- parity claims are only meaningful for the HIP path.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MJX_LIB") or os.path.join(_HERE, "csrc", "libmjx.so")   # MJX_LIB: A/B builds (tools/)

c_void_p, c_int, c_int64, c_float, c_double = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double

ALLREDUCE_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_int64, c_void_p)
REDUCE_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_int64, c_int, c_void_p)      # mjx_reduce_fn

# name -> (restype, argtypes); every symbol include/mjx.h declares
PROTOTYPES = {
    "mjx_last_error": (ctypes.c_char_p, []),
    "mjx_version": (c_int, []),
    "mjx_device_count": (c_int, []),
    "mjx_process_state": (c_int, [ctypes.POINTER(c_int64)]),
    "mjx_create": (c_int, [ctypes.POINTER(c_void_p), c_int, c_int, c_int, ctypes.POINTER(c_int), c_int]),
    "mjx_destroy": (None, [c_void_p]),
    "mjx_num_params": (c_int64, [c_void_p]),
    "mjx_uses_fused_path": (c_int, [c_void_p]),
    "mjx_malloc": (c_int, [ctypes.POINTER(c_void_p), c_int64]),
    "mjx_free": (c_int, [c_void_p]),
    "mjx_memcpy_h2d": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "mjx_memcpy_d2h": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "mjx_stream_sync": (c_int, [c_void_p]),
    "mjx_bind_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64]),
    "mjx_bind_rows": (c_int, [c_void_p, c_int64, c_int64, c_void_p]),
    "mjx_bind_policy": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
    "mjx_surr_vpg": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "mjx_fvp": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "mjx_eval_surr_kl": (c_int, [c_void_p, c_void_p, c_void_p]),
    "mjx_cg_solve": (c_int, [c_void_p, c_void_p, c_int, c_float, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),   # (the mjx_allreduce_fn argument as a plain pointer: ctypes.cast(ALLREDUCE_FN(f), c_void_p) or None)
    "mjx_cg_init": (c_int, [c_void_p, c_void_p, c_void_p]),
    "mjx_cg_p": (c_void_p, [c_void_p]),
    "mjx_cg_step": (c_int, [c_void_p, c_void_p, c_float, c_double, c_void_p]),
    "mjx_cg_finish": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mjx_comm_unique_id": (c_int, [ctypes.c_char_p]),
    "mjx_comm_init": (c_int, [c_void_p, c_int, c_int, ctypes.c_char_p]),
    "mjx_comm_destroy": (c_int, [c_void_p]),
    "mjx_comm_world": (c_int, [c_void_p]),
    "mjx_comm_set_callback": (c_int, [c_void_p, REDUCE_FN, c_void_p, c_int]),
    "mjx_comm_allreduce": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "mjx_peer_export": (c_int, [c_void_p, c_int, c_int, ctypes.c_char_p]),
    "mjx_peer_connect": (c_int, [c_void_p, ctypes.c_char_p]),
    "mjx_peer_status": (c_int, [c_void_p, ctypes.POINTER(c_int)]),
    "mjx_npg_update": (c_int, [c_void_p, c_int, c_float, c_double, c_double, c_double, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_void_p]),
    "mjx_trpo_update": (c_int, [c_void_p, c_int, c_float, c_double, c_double, c_double, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p]),
    "mjx_dapg_update": (c_int, [c_void_p, c_int, c_float, c_double, c_double, c_float, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p]),
    "mjx_apply_step": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p]),
    "mjx_apply_npg_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_float, c_void_p, c_void_p, c_void_p]),
    "mjx_time_index": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "mjx_discount_scan": (c_int, [c_void_p, c_void_p, c_int64, c_double, c_void_p, c_void_p]),
    "mjx_gae": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_double, c_double, c_void_p, c_void_p]),
    "mjx_sum_stats": (c_int, [c_void_p, c_int64, c_double, c_void_p, c_void_p]),
    "mjx_whiten_cast": (c_int, [c_void_p, c_int64, c_double, c_double, c_double, c_void_p, c_void_p]),
    "mjx_cast_f64_f32": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "mjx_policy_forward": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mjx_policy_minibatch_adam": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_float, c_float, c_void_p, c_void_p]),
    "mjx_host_gather": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int]),
    "mjx_host_gather_f64_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int]),
    "mjx_host_segment_sums": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int]),
    "mjx_host_mt19937_permutation": (c_int, [c_void_p, ctypes.POINTER(ctypes.c_int32), c_int64, c_void_p]),
    "mjx_host_mt19937_permutations": (c_int, [c_void_p, ctypes.POINTER(ctypes.c_int32), c_int64, c_int, c_void_p]),
    "mjx_host_mt19937_randint": (c_int, [c_void_p, ctypes.POINTER(ctypes.c_int32), c_int64, c_int64, c_void_p]),
    "mjx_stage_async": (c_int, [ctypes.POINTER(c_void_p), c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                c_int64, c_int, c_int, c_void_p]),
    "mjx_stage_wait": (c_int, [c_void_p]),
    "mjx_bl_num_features": (c_int, [c_int, c_int]),
    "mjx_bl_features_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "mjx_bl_gram": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "mjx_bl_predict": (c_int, [c_int, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "mjx_mlp_predict": (c_int, [c_void_p, c_int64, c_int, ctypes.POINTER(c_int), c_int, c_void_p, c_void_p, c_void_p]),
    "mjx_mlp_fit_adam": (c_int, [c_void_p, c_void_p, c_int64, c_int, ctypes.POINTER(c_int), c_int, c_void_p, c_void_p, c_void_p,
                                 c_int64, c_void_p, c_int, c_int, c_float, c_float, c_void_p, c_void_p]),
    "mjx_profile_enable": (c_int, [c_void_p, c_int]),
    "mjx_profile_read": (c_int, [c_void_p, ctypes.POINTER(c_double)]),
    "mjx_profile_samples": (c_int, [c_void_p, ctypes.POINTER(c_double), c_int, ctypes.POINTER(c_int)]),
    "mjx_set_debug_buffer": (c_int, [c_void_p, c_void_p, c_int64]),
    "mjx_set_clock_buffer": (c_int, [c_void_p, c_void_p]),
}

_lib = None


class MjxError(RuntimeError):
    pass


def load():
    """dlopen libmjx.so and attach prototypes (idempotent)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: its wheel bundles its own HIP runtime (libamdhip64.so.7 + HSA).  If libmjx.so is loaded -- and HIP initialised --
    # before torch, the process binds /opt/rocm's runtime instead and torch then reports "No HIP GPUs are available" (seen with
    # build() followed by smoke() in one process).  Loaded in this order the dynamic linker hands libmjx the runtime torch uses.
    try:
        import torch  # noqa: F401
    except Exception:       # pragma: no cover  (a host without torch: the C ABI alone)
        pass
    if not os.path.exists(LIB_PATH):
        raise MjxError("libmjx.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().mjx_last_error()
        raise MjxError("libmjx error %d: %s" % (rc, msg.decode() if msg else "?"))


def ptr(t):
    """device (or host) pointer of a torch tensor / None."""
    return None if t is None else c_void_p(t.data_ptr())
