"""Kind table of SPEC-COVIS.

Kind names are the ones the reference's consumers hard-code in file names
(``src/covisitation/inference.py:87-111``); the integer type-weight vectors are the
ones that literally occur in the reference.
"""
Q16 = 65536

# W = 65536 * sum over sessions of Wk[type_y]
TYPE_WEIGHTS = {
    'click_weighted': (1, 6, 3),   # src/baseline/aid_weight.py:34
    'cart_weighted': (1, 9, 6),    # src/covisitation/inference.py:72
    'order_weighted': (1, 3, 6),   # src/baseline/aid_weight.py:82
}
# M[type_x][type_y]; W = 65536 * number of sessions holding a masked pair
FILTER_MASKS = {
    'click_cart': ((0, 1, 0), (0, 0, 0), (0, 0, 0)),
    'click_order': ((0, 0, 1), (0, 0, 0), (0, 0, 0)),
    'cart_order': ((0, 0, 0), (0, 1, 1), (0, 1, 1)),
    'click_click': ((1, 0, 0), (0, 0, 0), (0, 0, 0)),   # BASELINE.json config 1
}
TIME_KIND = 'time_weighted'
REFERENCE_KINDS = ('time_weighted', 'click_weighted', 'cart_weighted', 'order_weighted',
                   'click_cart', 'click_order', 'cart_order')
ALL_KINDS = REFERENCE_KINDS + ('click_click',)


def mask_bits(mask):
    """3x3 mask -> 9-bit word, bit (type_x * 3 + type_y)."""
    bits = 0
    for tx in range(3):
        for ty in range(3):
            if mask[tx][ty]:
                bits |= 1 << (tx * 3 + ty)
    return bits
