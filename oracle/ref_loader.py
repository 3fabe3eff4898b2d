"""Import the UNMODIFIED reference (aravindr93/mjrl) without gym / mujoco_py -- TEST / MEASUREMENT INFRASTRUCTURE.

Two places it can come from: the sources at /root/reference (build container) or the bytecode oracle/ref_stage.py compiled from
them into oracle/_ref/ (what travels to the GPU box).  Same three stubs as tests/golden/_ref_import.py: the package __init__ is
skipped (it imports gym + mujoco_py), ``gym`` is a two-attribute stand-in, ``mjrl.samplers.batch_sampler`` (imported by trpo.py:15,
absent from the tree) is an empty module.  Used by bench.py's ``cpu_baseline`` leg and by tests; never by the product.
"""
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
STAGED = os.path.join(HERE, "_ref")


def root():
    """directory that holds the ``mjrl`` package to import, or None"""
    if os.path.isdir("/root/reference/mjrl") and os.environ.get("MJX_REF_STAGED_ONLY") != "1":
        return "/root/reference"
    if os.path.exists(os.path.join(STAGED, "mjrl", "algos", "npg_cg.pyc")):
        return STAGED
    return None


def example_script(name="policy_opt_job_script"):
    """path of one of the reference's example scripts: the source in the build container, the staged bytecode elsewhere (or None)"""
    r = root()
    if r is None:
        return None
    for p in (os.path.join(r, "examples", name + ".py"), os.path.join(r, "examples", name + ".pyc")):
        if os.path.exists(p):
            return p
    return None


def install():
    """-> the root the reference was bound to, or None when it is not available here"""
    r = root()
    if r is None:
        return None
    pkg = sys.modules.get("mjrl")
    if pkg is not None and getattr(pkg, "__ref_stub__", False):
        return getattr(pkg, "__ref_root__", r)          # (tests/golden/_ref_import.py installs the same stubs without the tag)
    sys.dont_write_bytecode = True
    pkg = types.ModuleType("mjrl")
    pkg.__path__ = [os.path.join(r, "mjrl")]          # skip mjrl/__init__.py (imports gym + mujoco_py)
    pkg.__ref_stub__, pkg.__ref_root__ = True, r
    sys.modules["mjrl"] = pkg
    gym = types.ModuleType("gym")
    gym.Env = type("Env", (), {})
    gym.make = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("no gym here"))
    sys.modules.setdefault("gym", gym)
    sys.modules.setdefault("mjrl.samplers.batch_sampler", types.ModuleType("mjrl.samplers.batch_sampler"))
    sys.modules.setdefault("mjrl.envs", types.ModuleType("mjrl.envs"))       # (`import mjrl.envs` of the job scripts: gym registrations + mujoco_py)
    return r
