"""Covisitation-matrix builder (the component the reference lacks)
behind the reference's script + parquet contract. See ``builder.py`` (CLI) and
``engine.py`` (device engine over the C-ABI)."""
from .spec import ALL_KINDS, REFERENCE_KINDS, TYPE_WEIGHTS, FILTER_MASKS, Q16  # noqa: F401
