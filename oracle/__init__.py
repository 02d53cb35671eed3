"""ORACLE — test infrastructure only.

CPU restatement of the SGAM per-step generative-sensing hot path,
used exclusively as the parity checker by ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py``.  The product package
(``sgam_neurips22_amd``) never imports anything from here.

Pinning: every function below is checked in ``tests/test_oracle_golden.py`` against
golden vectors produced by importing the reference itself in the build container
(``tests/golden/gen_golden.py``); the warp functions bit-for-bit, the VQGAN
functions to 2e-5 absolute (both sides are fp32 torch-CPU, differing only in
summation order inside conv/bmm).
"""
