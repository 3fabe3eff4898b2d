"""mjrl.baselines.linear_baseline.LinearBaseline drop-in (reference linear_baseline.py:5-65)."""
from .quadratic_baseline import LinearBaseline  # noqa: F401
