"""MI355X matrix-factorization trainer / scorer behind the reference's
``src/matrix_factorization`` module surface (``torch_modules``, ``torch_trainer``,
``torch_utils``, ``metrics``) plus the BPR trainer and full-sort scorer the
reference reaches through recbole (``src/recbole``)."""
