"""ctypes binding of ``csrc/libotto_amd.so`` (the C-ABI declared in ``include/*.h``).

The product path has NO CPU fallback: if the HIP library is missing or a symbol is
absent this module raises at first use, and every entry point raises
``OttoError`` with ``otto_last_error()`` on a non-zero return code.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# OTTO_AMD_LIB selects another build of the SAME library (the phase-profiling build of tools/perf_covis.py --prof)
LIB_PATH = os.environ.get('OTTO_AMD_LIB') or os.path.join(_HERE, 'csrc', 'libotto_amd.so')

MAX_FILTERS = 4
MAX_TYPE_WEIGHTS = 4
GROUP_TYPE, GROUP_FILTER, GROUP_TIME = 0, 1, 2
STAT_NAMES = ('sessions', 'tail_events', 'pair_slots', 'pairs', 'runs', 'items_s', 'items_m', 'items_l', 'retries',
              'pairs_s', 'pairs_m', 'pairs_l', 'runs_s', 'runs_m', 'runs_l', 'shared_runs', 'row_records')
TIMING_NAMES = ('winscan', 'expand', 'index', 'partition', 'reduce_s', 'reduce_m', 'reduce_l', 'merge')


class OttoError(RuntimeError):
    pass


class CandParams(C.Structure):
    # mirrors otto_cand_params (include/otto_cand.h)
    _fields_ = [
        ('n_aids', C.c_uint32),
        ('k', C.c_int32),
        ('n_matrices', C.c_int32),
        ('d_mat_y', C.c_void_p * 8),
        ('d_mat_n', C.c_void_p * 8),
        ('n_terms', C.c_int32),
        ('term_matrix', C.c_int32 * 8),
        ('term_source', C.c_int32 * 8),
        ('n_common', C.c_int32),
        ('mat_k', C.c_int32 * 8),
    ]


class RecencyParams(C.Structure):
    # mirrors otto_recency_params (include/otto_cand.h)
    _fields_ = [
        ('n_curves', C.c_int32),
        ('start', C.c_double * 4),
        ('stop', C.c_double * 4),
        ('type_coef', C.c_double * 3),
    ]


class RecencyPredParams(C.Structure):
    # mirrors otto_recency_pred_params (include/otto_cand.h)
    _fields_ = [
        ('n_targets', C.c_int32),
        ('start', C.c_double * 3),
        ('stop', C.c_double * 3),
        ('bump', C.c_double * 3),
        ('type_coef', C.c_double * 3),
        ('n_common', C.c_int32),
        ('n_pred', C.c_int32),
        ('min_unique', C.c_int32),
        ('d_cand', C.c_void_p * 3),
        ('d_count', C.c_void_p * 3),
        ('d_n_cand', C.c_void_p * 3),
        ('d_self_count', C.c_void_p * 3),
    ]


class CovisParams(C.Structure):
    # mirrors otto_covis_params (include/otto_covis.h)
    _fields_ = [
        ('window', C.c_int32),
        ('max_gap', C.c_int32),
        ('n_aids', C.c_uint32),
        ('ts_min', C.c_int32),
        ('ts_max', C.c_int32),
        ('want_time', C.c_int32),
        ('n_filters', C.c_int32),
        ('filter_mask', C.c_uint16 * MAX_FILTERS),
        ('n_type_weights', C.c_int32),
        ('type_weight', (C.c_int32 * 3) * MAX_TYPE_WEIGHTS),
    ]


_vp, _i64, _i32, _u32 = C.c_void_p, C.c_int64, C.c_int, C.c_uint32
_p_i64 = C.POINTER(C.c_int64)

# every symbol include/*.h declares: name -> (restype, argtypes)
SIGNATURES = {
    # include/otto_covis.h
    'otto_last_error': (C.c_char_p, []),
    'otto_covis_create': (_i32, [C.POINTER(_vp), C.POINTER(CovisParams)]),
    'otto_covis_destroy': (None, [_vp]),
    'otto_covis_reset': (_i32, [_vp]),
    'otto_covis_feed': (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    'otto_covis_finalize': (_i32, [_vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    'otto_covis_stats': (_i32, [_vp, _p_i64]),
    'otto_covis_set_option': (_i32, [_vp, C.c_char_p, _i64]),
    'otto_covis_export_count': (_i32, [_vp, _u32, _u32, _p_i64, _p_i64, _vp]),
    'otto_covis_export_runs': (_i32, [_vp, _u32, _u32, _vp, _vp, _vp, _vp]),
    'otto_covis_import_runs': (_i32, [_vp, _vp, _i64, _vp, _vp, _i64, _vp]),
    'otto_covis_import_reserve': (_i32, [_vp, _i64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _vp]),
    'otto_covis_export_plan': (_i32, [_vp, _i32, _vp, _p_i64, _p_i64, _vp]),
    'otto_covis_export_fill': (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    'otto_covis_export_plan_range': (_i32, [_vp, _i32, _vp, _i64, _i64, _p_i64, _p_i64, _vp]),
    'otto_covis_export_fill_range': (_i32, [_vp, _i32, _vp, _i64, _i64, _p_i64, _p_i64, _vp, _vp, _vp, _vp]),
    'otto_covis_copy_records': (_i32, [_vp, _vp, _vp, _vp, _vp]),
    'otto_covis_timings': (_i32, [_vp, C.POINTER(C.c_float)]),
    'otto_covis_kernel_names': (_i32, [_vp, _i32, C.c_char_p, _i32]),
    'otto_debug_calibrate': (_i32, [_vp, _i64, _i32, _vp]),
    # include/otto_cand.h
    'otto_cand_lookup': (_i32, [C.POINTER(CandParams), _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    'otto_cand_lookup_self': (_i32, [C.POINTER(CandParams), _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    'otto_recency_predictions': (_i32, [C.POINTER(RecencyPredParams), _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    'otto_cand_predictions': (_i32, [_vp, _vp, _i64, _vp, _vp, _i32, _vp, _i32, _i32, _vp, _vp, _vp]),
    'otto_cand_ranker_workspace': (_i64, [_i64]),
    'otto_cand_ranker_rows': (_i32, [_vp, _vp, _i64, _vp, _i32, _vp, _p_i64, _vp, _i64, _vp]),
    'otto_cand_ranker_table': (_i32, [_vp, _vp, _i64, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'otto_recency_candidates': (_i32, [C.POINTER(RecencyParams), _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp]),
    # include/otto_events.h
    'otto_events_sort_workspace': (_i64, [_i64]),
    'otto_events_sort': (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _p_i64, _vp, _i64, _vp]),
    'otto_events_type_from_strings': (_i32, [_vp, _i32, _vp, _i64, _vp, _vp]),
    # include/otto_pairs.h
    'otto_pairs_raw_count': (_i32, [_vp, _i64, _i32, _p_i64, _vp]),
    'otto_pairs_workspace': (_i64, [_i64]),
    'otto_pairs_time': (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp, _vp, _vp, _p_i64, _vp, _i64, _vp]),
    'otto_pairs_diff': (_i32, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _p_i64, _vp, _i64, _vp]),
    # include/otto_inter.h
    'otto_inter_workspace': (_i64, [_u32]),
    'otto_inter_features': (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _i32, _u32, _vp, _vp, _vp, _vp, _i64, _vp]),
    'otto_inter_features_rows': (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _i64, _vp]),
    # include/otto_mf.h
    'otto_mf_create': (_i32, [C.POINTER(_vp), _i64, _i64, _i32, _i64, _i32]),
    'otto_mf_destroy': (None, [_vp]),
    'otto_mf_forward': (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    'otto_mf_eval': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp]),
    'otto_mf_eval_sums': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp]),
    'otto_mf_read_sums': (_i32, [_vp, C.POINTER(C.c_double), _i32, _vp]),
    'otto_mf_check': (_i32, [_vp, _p_i64, _vp]),
    'otto_mf_step_sparse_adam': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32,
                                        C.c_double, C.c_double, C.c_double, C.c_double, _i64, _vp, _vp]),
    'otto_mf_bpr_step': (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, C.c_uint64, C.c_uint64, _i64, C.c_float, C.c_float,
                                _i32, _vp, _vp, _vp]),
    'otto_mf_score_topk': (_i32, [_vp, _vp, _i64, _i64, _i32, _i32, _i64, _vp, _vp, _vp, _i64, _vp]),
    'otto_mf_score_workspace': (_i64, [_i64, _i64, _i32]),
    'otto_mf_topk_merge': (_i32, [_vp, _vp, _i32, _i64, _i32, _vp, _vp, _vp]),
}

_lib = None


def register(signatures):
    """Let sibling modules (matrix_factorization) add their header's symbols."""
    SIGNATURES.update(signatures)
    global _lib
    if _lib is not None:
        _bind(_lib, signatures)


def _bind(lib, signatures):
    for name, (res, args) in signatures.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise OttoError(f'{LIB_PATH} does not export {name}; rebuild with __graft_entry__.build()') from e
        fn.restype = res
        fn.argtypes = args


def lib():
    """Load (once) and return the shared library; raises OttoError when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OttoError(
                f'HIP library {LIB_PATH} is not built. Run `python -c "import __graft_entry__ as g; g.build()"` '
                f'(or `make -C {os.path.dirname(LIB_PATH)}`). There is no CPU fallback.')
        try:
            loaded = C.CDLL(LIB_PATH)
        except OSError as e:
            raise OttoError(f'cannot load {LIB_PATH}: {e}') from e
        _bind(loaded, SIGNATURES)
        _lib = loaded
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().otto_last_error()
        raise OttoError(f'{what} failed (code {rc}): {msg.decode() if msg else "?"}')
