"""Run an UNMODIFIED mjrl script on the MI355X path:   python -m mjrl_amd.dropin examples/policy_opt_job_script.py --output job --config cfg.txt

mjrl's scripts pick their classes with import statements (``from mjrl.policies.gaussian_mlp import MLP``,
``from mjrl.algos.npg_cg import NPG`` ... -- examples/policy_opt_job_script.py:8-15).  ``install()`` binds those module names to
this package's modules in ``sys.modules`` BEFORE the script runs, so the same script -- byte for byte -- constructs the GPU
policy / baselines / agents and hands them to mjrl's own ``train_agent`` and samplers, which stay what they are
(``mjrl.utils.*``, ``mjrl.samplers.*``, ``mjrl.envs`` are not touched).  Editing the import block by hand
does the same thing; this is the zero-edit form.
"""
import importlib
import runpy
import sys

# reference module -> the module of this package that exports the same class names with the same constructor arguments
ALIASES = {
    "mjrl.policies.gaussian_mlp": "mjrl_amd.policies.gaussian_mlp",          # MLP
    "mjrl.policies.gaussian_linear": "mjrl_amd.policies.gaussian_linear",    # LinearPolicy
    "mjrl.baselines.quadratic_baseline": "mjrl_amd.baselines.quadratic_baseline",
    "mjrl.baselines.linear_baseline": "mjrl_amd.baselines.linear_baseline",
    "mjrl.baselines.mlp_baseline": "mjrl_amd.baselines.mlp_baseline",
    "mjrl.baselines.zero_baseline": "mjrl_amd.baselines.zero_baseline",
    "mjrl.algos.batch_reinforce": "mjrl_amd.algos.batch_reinforce",          # BatchREINFORCE (VPG / NVPG)
    "mjrl.algos.npg_cg": "mjrl_amd.algos.npg_cg",                            # NPG
    "mjrl.algos.trpo": "mjrl_amd.algos.trpo",
    "mjrl.algos.dapg": "mjrl_amd.algos.dapg",
    "mjrl.algos.ppo_clip": "mjrl_amd.algos.ppo_clip",
    "mjrl.algos.behavior_cloning": "mjrl_amd.algos.behavior_cloning",
    "mjrl.utils.process_samples": "mjrl_amd.utils.process_samples",
}


def install(verbose=False):
    """bind the reference's class-bearing module names to this package's modules -> the list of names bound.  mjrl itself
    must be importable (its utils / samplers / envs are used as they are)."""
    import mjrl  # noqa: F401  (the package the script's other imports come from; ImportError here is the honest failure)
    bound = []
    for ref_name, ours in ALIASES.items():
        mod = importlib.import_module(ours)
        sys.modules[ref_name] = mod
        parent, _, leaf = ref_name.rpartition(".")
        if parent in sys.modules:                       # `import mjrl.algos.npg_cg as x` resolves through the parent's attribute
            try:
                setattr(sys.modules[parent], leaf, mod)
            except Exception:                           # pragma: no cover
                pass
        bound.append(ref_name)
    if verbose:
        print("[mjrl_amd.dropin] %d mjrl modules bound to the MI355X path" % len(bound), file=sys.stderr)
    return bound


def run(script, argv=()):
    """install() and run `script` (a .py or compiled .pyc file) as __main__ with sys.argv = [script, *argv]"""
    install()
    old = sys.argv
    sys.argv = [script] + list(argv)
    try:
        return runpy.run_path(script, run_name="__main__")
    finally:
        sys.argv = old


def main():
    if len(sys.argv) < 2:
        sys.exit("usage: python -m mjrl_amd.dropin <mjrl script.py> [its arguments ...]")
    run(sys.argv[1], sys.argv[2:])


if __name__ == "__main__":
    main()
