"""CPU oracle for the mjrl NPG/TRPO update path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``mjrl_amd/`` may import this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it,
and only as the checker / the timed CPU baseline, never as the product path.

Parity status: PINNED.  The reference (aravindr93/mjrl) ships no golden
vectors, so ``tests/golden/make_golden.py`` imports the *unmodified* reference
from /root/reference in the build container, runs its own
``flat_vpg / HVP / cg_solve / train_from_paths / compute_advantages / fit``
on seeded synthetic paths and stores inputs' seeds + outputs as ``.npz``
fixtures.  ``tests/test_oracle_golden.py`` checks this oracle against those
fixtures, and the GPU tests check the HIP path against both.
"""
