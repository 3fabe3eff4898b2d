"""mjrl_amd -- MI355X-native NPG / TRPO / DAPG update path behind mjrl's operator surface.

Host side (this package) mirrors the reference's Policy / Baseline / Agent classes;
all batch arithmetic runs in libmjx.so (mjrl_amd/csrc, C ABI in include/mjx.h).
"""
__version__ = "0.1.0"
