"""CPU oracle for the MLVectorDB brute-force kNN hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``mlvectordb_amd/`` may import this package: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it, and only as
the checker / the timed CPU baseline, never as the product path.

Pinning status (see DESIGN.md "Oracle"):
  * The reference's search arithmetic lives in hnswlib==0.8.0 (reference
    pyproject.toml:12, poetry.lock:144-145), which is NOT in /root/reference and not
    installable here; its published distance spaces are restated in
    ``oracle/exact_scan.py``.
  * The oracle is checked against every known-answer test the reference holds for this
    path (reference tests/test_index.py, tests/test_query_processor.py: ordering,
    membership, counts, sign, type) -- see tests/test_reference_behaviour.py.
  * The reference holds no golden vectors and no test asserts a numeric score value:
    for numeric scores this oracle is **parity unpinned**.
"""
from .exact_scan import (  # noqa: F401
    SPACES,
    exact_distances,
    knn,
    range_query,
    postprocess_score,
)
