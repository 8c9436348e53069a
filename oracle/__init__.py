"""TEST INFRASTRUCTURE ONLY — CPU restatement of the FedFR hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and there only as the checker / reported baseline.  The product path
(``fedfr_amd``) never imports this package and fails loudly when the HIP library
is missing.

Parity pin: the restatement is checked against golden vectors captured from the
imported reference (``tools/make_golden.py`` → ``tests/golden/*.npz``); see
``tests/test_oracle_golden.py``.
"""
