"""Import the UNMODIFIED reference (aravindr93/mjrl at /root/reference) without
gym / mujoco_py.  Only used by make_golden.py in the build
container; never on the GPU box, never by the product."""
import sys
import types

REF = "/root/reference"


def install():
    sys.dont_write_bytecode = True
    if "mjrl" in sys.modules and getattr(sys.modules["mjrl"], "__ref_stub__", False):
        return
    pkg = types.ModuleType("mjrl")
    pkg.__path__ = [REF + "/mjrl"]          # skip mjrl/__init__.py (imports gym + mujoco_py)
    pkg.__ref_stub__ = True
    sys.modules["mjrl"] = pkg
    gym = types.ModuleType("gym")
    gym.Env = type("Env", (), {})
    gym.make = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("no gym here"))
    sys.modules.setdefault("gym", gym)
    # trpo.py:15 imports a module that does not exist in the tree
    sys.modules.setdefault("mjrl.samplers.batch_sampler", types.ModuleType("mjrl.samplers.batch_sampler"))
