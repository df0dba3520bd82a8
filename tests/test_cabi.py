"""The C-ABI library builds, loads, and exports every symbol include/*.h declares; argument
validation works without touching a GPU (no compute calls here)."""
import ctypes as C
import glob
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, 'include', '*.h')):
        txt = open(h).read()
        txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
        names |= set(re.findall(r'\b(otto_[a-z0-9_]+)\s*\(', txt))
    return names


@pytest.fixture(scope='module')
def lib():
    import __graft_entry__ as g
    g.build()
    from otto_amd import _lib
    return _lib.lib()


def test_every_declared_symbol_is_exported_and_bound(lib):
    from otto_amd import _lib
    declared = _declared_symbols()
    assert declared, 'no declarations found in include/*.h'
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in include/ but not exported'
    assert declared <= set(_lib.SIGNATURES), f'unbound: {declared - set(_lib.SIGNATURES)}'


def test_params_struct_matches_header_layout(lib):
    from otto_amd import _lib
    # window, max_gap, n_aids, ts_min, ts_max, want_time, n_filters (7 x 4) + 4 x u16 + n_type_weights + 4x3 i32
    assert C.sizeof(_lib.CovisParams) == 7 * 4 + 4 * 2 + 4 + 48


def test_create_rejects_bad_params_with_message(lib):
    from otto_amd import _lib
    ctx = C.c_void_p()
    p = _lib.CovisParams()
    p.window, p.max_gap, p.n_aids = 64, 10, 100
    rc = lib.otto_covis_create(C.byref(ctx), C.byref(p))
    assert rc == -22 and b'window' in lib.otto_last_error()
    p.window, p.n_aids = 30, 0
    assert lib.otto_covis_create(C.byref(ctx), C.byref(p)) == -22 and b'n_aids' in lib.otto_last_error()
    p.n_aids, p.n_type_weights = 100, 1
    assert lib.otto_covis_create(C.byref(ctx), C.byref(p)) == -22 and b'type_weight' in lib.otto_last_error()
    with pytest.raises(_lib.OttoError):
        _lib.check(-22, 'otto_covis_create')


def test_engine_refuses_cpu_device():
    from otto_amd import _lib
    from otto_amd.covisitation.engine import CovisBuilder
    with pytest.raises(_lib.OttoError):
        CovisBuilder(100, device='cpu')


def test_missing_library_fails_loudly(monkeypatch):
    from otto_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libotto_amd.so')
    with pytest.raises(_lib.OttoError, match='no CPU fallback'):
        _lib.lib()
