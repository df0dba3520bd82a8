"""MI355X-native covisitation builder + matrix-factorization trainer/scorer.

Drop-in for the hot path of gunesevitan/otto-multi-objective-recommender-system:
the (missing) covisitation-matrix builder that feeds ``src/covisitation/inference.py``
and ``src/ranker/*candidate_generation.py``, and ``src/matrix_factorization``.
Host code is Python over a C-ABI shared library (``csrc/libotto_amd.so``) of
hand-written gfx950 HIP kernels; see DESIGN.md / INTEGRATION.md.

Imported as ``otto_amd`` (the directory name is not a Python identifier).
"""

__version__ = '0.1.0'
