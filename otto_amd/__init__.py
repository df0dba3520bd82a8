"""Import alias for the package directory ``otto-multi-objective-recommender-system_amd/``.

The product package lives in a directory whose name is not a valid Python
identifier, so ``import otto_amd`` resolves its submodules from that directory
(``otto_amd.covisitation`` -> ``otto-multi-objective-recommender-system_amd/covisitation``).
No code lives here.
"""
import os as _os

_PKG_DIR = _os.path.join(
    _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
    'otto-multi-objective-recommender-system_amd',
)
__path__.insert(0, _PKG_DIR)

# Run the real package's __init__ in this namespace.
with open(_os.path.join(_PKG_DIR, '__init__.py')) as _f:
    exec(compile(_f.read(), _os.path.join(_PKG_DIR, '__init__.py'), 'exec'))
