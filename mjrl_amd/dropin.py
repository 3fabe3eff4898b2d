"""Run an UNMODIFIED mjrl script on the MI355X path:   python -m mjrl_amd.dropin examples/policy_opt_job_script.py --output job --config cfg.txt

mjrl's scripts pick their classes with import statements (``from mjrl.policies.gaussian_mlp import MLP``,
``from mjrl.algos.npg_cg import NPG`` ... -- examples/policy_opt_job_script.py:8-15).  ``install()`` binds those module names to
this package's modules in ``sys.modules`` BEFORE the script runs, so the same script -- byte for byte -- constructs the GPU
policy / baselines / agents and hands them to mjrl's own ``train_agent`` and samplers, which stay what they are
(``mjrl.utils.*``, ``mjrl.samplers.*``, ``mjrl.envs`` are not touched).  Editing the import block by hand
does the same thing; this is the zero-edit form.
"""
import runpy
import sys

def _aliases():
    """reference module name -> the module of this package that exports the same class names with the same constructor arguments"""
    from .algos import batch_reinforce, behavior_cloning, dapg, npg_cg, ppo_clip, trpo
    from .baselines import linear_baseline, mlp_baseline, quadratic_baseline, zero_baseline
    from .policies import gaussian_linear, gaussian_mlp
    from .utils import process_samples
    return {
        "mjrl.policies.gaussian_mlp": gaussian_mlp,                  # MLP
        "mjrl.policies.gaussian_linear": gaussian_linear,            # LinearPolicy
        "mjrl.baselines.quadratic_baseline": quadratic_baseline,
        "mjrl.baselines.linear_baseline": linear_baseline,
        "mjrl.baselines.mlp_baseline": mlp_baseline,
        "mjrl.baselines.zero_baseline": zero_baseline,
        "mjrl.algos.batch_reinforce": batch_reinforce,               # BatchREINFORCE (VPG / NVPG)
        "mjrl.algos.npg_cg": npg_cg,                                 # NPG
        "mjrl.algos.trpo": trpo,
        "mjrl.algos.dapg": dapg,
        "mjrl.algos.ppo_clip": ppo_clip,
        "mjrl.algos.behavior_cloning": behavior_cloning,
        "mjrl.utils.process_samples": process_samples,
    }


def install(verbose=False):
    """bind the reference's class-bearing module names to this package's modules -> the list of names bound.  mjrl itself
    must be importable (its utils / samplers / envs are used as they are)."""
    import mjrl  # noqa: F401  (the package the script's other imports come from; ImportError here is the honest failure)
    import mjrl.algos  # noqa: F401  (the parent packages first: `import mjrl.algos.npg_cg as x` binds through their attributes)
    import mjrl.baselines  # noqa: F401
    import mjrl.policies  # noqa: F401
    import mjrl.utils  # noqa: F401
    bound = []
    for ref_name, mod in _aliases().items():
        sys.modules[ref_name] = mod
        parent, _, leaf = ref_name.rpartition(".")
        if parent in sys.modules:                       # `import mjrl.algos.npg_cg as x` resolves through the parent's attribute
            try:
                setattr(sys.modules[parent], leaf, mod)
            except Exception:                           # pragma: no cover
                pass
        bound.append(ref_name)
    from .utils import ingest
    ingest.tune_malloc()                                # a training process from here on (utils/ingest.py; MJX_MALLOC_TUNE=0 opts out)
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
