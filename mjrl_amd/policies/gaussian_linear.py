"""mjrl.policies.gaussian_linear.LinearPolicy drop-in (reference gaussian_linear.py:9-139)."""
from .gaussian_mlp import LinearPolicy  # noqa: F401
