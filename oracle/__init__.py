"""CPU oracle for the deepinv hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.  ``deepinv_amd/`` never does (checked by tests/test_layout.py).

Two layers:
  * ``oracle.naive``       – fp64 numpy restatements straight from the defining sums
                             (direct DFT, ray sums …); O(N^2), small sizes only.
  * ``oracle.physics_cpu`` – restatement of the reference's own algorithm: the same ATen CPU
    / ``oracle.optim_cpu``   call sequence (torch.fft, grid_sample, conv2d …) as plain functions,
    / ``oracle.drunet_cpu``  each citing the reference file:line it follows.  Fast enough for
                             the full BASELINE configs; also timed as the ``cpu_baseline``
                             ("port") on the GPU box where /root/reference does not exist.

Pinning: ``tests/golden/*.npz`` hold input/output vectors generated from the *real* reference
imported through ``oracle.ref_shim`` (script: tests/golden/make_golden.py, committed), plus the
reference's literal doctest vectors; tests/test_oracle_golden.py checks both oracle layers
against them, so parity is pinned (not "parity unpinned").
"""
