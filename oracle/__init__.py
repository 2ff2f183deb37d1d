"""CPU oracle package (TEST INFRASTRUCTURE ONLY — see oracle/pn2_oracle.c).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package.  The product (``4d-or_amd/``) never does.
"""
