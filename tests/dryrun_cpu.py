"""Development aid (NOT a test, TEST INFRASTRUCTURE): run the Python-level logic of selected -m gpu tests in the
GPU-less build container by standing tests/_cpu_backend.OracleBackend in for mjrl_amd.engine.HipBackend, so that
typos / shape errors / wrong fixture keys are found before GPU minutes are spent.  Says nothing about the kernels.

    python tests/dryrun_cpu.py [test_function_name ...]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import mjrl_amd.engine as E  # noqa: E402
from tests._cpu_backend import OracleBackend  # noqa: E402


def _backend(n, m, hidden, device=None):
    return OracleBackend(n, m, hidden)


E.HipBackend = _backend

import tests.test_gpu_operators as T  # noqa: E402

DEFAULT = [
    ("test_agent_operator_methods_vs_reference", ("npg_cfg2_small",)),
    ("test_agent_operator_methods_vs_reference", ("npg_cfg1_linear",)),
    ("test_CG_solve_standalone_equals_cg_solve_and_reference", ()),
    ("test_const_learn_rate_branch", ("npg_pointmass_32x32",)),
    ("test_dapg_with_hvp_sample_frac_draws_from_the_on_policy_rows", ()),
]
# (test_one_call_update_equals_call_sequence needs the real library: the oracle stand-in has no mjx_npg_update)

if __name__ == "__main__":
    want = sys.argv[1:]
    for name, args in DEFAULT:
        if want and name not in want:
            continue
        print("dry run:", name, args, flush=True)
        getattr(T, name)(*args)
    print("ok")
