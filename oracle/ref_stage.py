"""Stage the UNMODIFIED reference's update path as compiled bytecode under oracle/_ref/ (TEST / MEASUREMENT INFRASTRUCTURE).

The reference (aravindr93/mjrl, /root/reference) is pure Python; it exists in the build container only.  bench.py's
``cpu_baseline`` wants to time the reference ITSELF -- ``mjrl.algos.npg_cg.NPG.train_from_paths`` (mjrl/algos/npg_cg.py:91-163)
-- on the GPU box's host cores.  This recipe byte-compiles the modules that call needs, from the sources where they lie, into
``oracle/_ref/mjrl/**.pyc`` (sourceless modules: CPython imports ``name.pyc`` from a package directory).  Only build OUTPUT goes
there; no reference source is copied anywhere.  ``oracle/_ref/`` is git-ignored (never in history) but travels to the GPU box
with the snapshot, like the built libmjx.so.  ``__graft_entry__.build()`` runs this whenever /root/reference is present.

Nothing under mjrl_amd/ imports this or the staged modules; only ``bench.py``'s ``cpu_baseline`` leg (and the tests) do, through
``oracle/ref_loader.py``.
"""
import os
import py_compile
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("MJX_REFERENCE_ROOT", "/root/reference")
OUT = os.path.join(HERE, "_ref")

# the modules behind NPG / TRPO / DAPG.train_from_paths, process_samples, the baselines and the policy (SURVEY 8c "files a CPU
# restatement must follow") plus what they import at module level (samplers.core -> utils.gym_env / tensor_utils; logger)
MODULES = [
    "algos/__init__", "algos/batch_reinforce", "algos/npg_cg", "algos/trpo", "algos/dapg", "algos/ppo_clip", "algos/behavior_cloning",
    "policies/__init__", "policies/gaussian_mlp", "policies/gaussian_linear",
    "baselines/__init__", "baselines/baseline", "baselines/linear_baseline", "baselines/quadratic_baseline", "baselines/mlp_baseline",
    "baselines/zero_baseline",
    "utils/__init__", "utils/fc_network", "utils/cg_solve", "utils/process_samples", "utils/optimize_model", "utils/logger",
    "utils/gym_env", "utils/tensor_utils", "utils/make_train_plots", "utils/train_agent",
    "samplers/__init__", "samplers/core",
]


EXAMPLES = ["examples/policy_opt_job_script"]


def stage(verbose=True):
    """-> number of modules compiled (0 when the reference is not present: the GPU box)"""
    src_root = os.path.join(REF, "mjrl")
    if not os.path.isdir(src_root):
        return 0
    n = 0
    for mod in MODULES:
        src = os.path.join(src_root, mod + ".py")
        if not os.path.exists(src):
            continue
        dst = os.path.join(OUT, "mjrl", mod + ".pyc")
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        if not os.path.exists(dst) or os.path.getmtime(dst) < os.path.getmtime(src):
            # dfile: what tracebacks name; UNCHECKED_HASH: the .pyc is valid on its own (no source to compare with)
            py_compile.compile(src, cfile=dst, dfile="<reference>/mjrl/%s.py" % mod, doraise=True,
                               invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
        n += 1
    # the job script the north star names (examples/policy_opt_job_script.py): run UNMODIFIED by tests/test_reference_driver.py
    # through mjrl_amd.dropin -- bytecode only, like the modules above
    for rel in EXAMPLES:
        src = os.path.join(REF, rel + ".py")
        if os.path.exists(src):
            dst = os.path.join(OUT, rel + ".pyc")
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            if not os.path.exists(dst) or os.path.getmtime(dst) < os.path.getmtime(src):
                py_compile.compile(src, cfile=dst, dfile="<reference>/%s.py" % rel, doraise=True,
                                   invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
            n += 1
    with open(os.path.join(OUT, "STAGED"), "w") as f:
        f.write("python %d.%d bytecode of %d modules of %s/mjrl (oracle/ref_stage.py); build output, not source\n"
                % (sys.version_info[0], sys.version_info[1], n, REF))
    if verbose:
        print("[ref_stage] %d reference modules byte-compiled into %s" % (n, OUT), flush=True)
    return n


if __name__ == "__main__":
    stage()
