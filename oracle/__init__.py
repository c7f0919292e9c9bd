"""CPU oracle for the MI355X MH-likelihood hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``mcmc_amd/`` may import this package:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` use it, and only as the checker - never as the thing shipped or
measured as the product.
"""
