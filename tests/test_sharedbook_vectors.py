"""The reference's own codebook self-test (lib/sharedbook.c:474-610, built as test_sharedbook): the only vectors
the reference holds that touch the encode path's tables.  Its books 4 and 5 are maptype-1 lattices
(q_min = -533200896, q_delta = 1611661312, 27 entries of dimension 3, quantlist {0, 7, 2}) with hand-written
dequantised values; books 2 and 3 carry the same packed floats.  The encode side derives {quantvals, minval, delta}
from exactly those header fields (vorbis_book_init_encode, lib/sharedbook.c:303-317): checked here for the
oracle (oracle/orc_book.c) and for the product's host code (csrc/setup_host.cpp), against the values the
reference's expected arrays imply."""
import ctypes as C

import numpy as np

Q_MIN, Q_DELTA = -533200896, 1611661312          # lib/sharedbook.c:513, :524, :535, :554
QUANTLIST = [0, 7, 2]                            # partial_quantlist1, :492
# test4_result (:539-547): entry e, component k -> quantlist[(e / 3^k) % 3] * delta + min, non-sequential
TEST4 = [-3, -3, -3, 4, -3, -3, -1, -3, -3, -3, 4, -3, 4, 4, -3, -1, 4, -3, -3, -1, -3, 4, -1, -3, -1, -1, -3,
         -3, -3, 4, 4, -3, 4, -1, -3, 4, -3, 4, 4, 4, 4, 4, -1, 4, 4, -3, -1, 4, 4, -1, 4, -1, -1, 4,
         -3, -3, -1, 4, -3, -1, -1, -3, -1, -3, 4, -1, 4, 4, -1, -1, 4, -1, -3, -1, -1, 4, -1, -1, -1, -1, -1]
# test5_result (:558-566): the same book with q_sequencep = 1 (running sums along the dimension)
TEST5 = [-3, -6, -9, 4, 1, -2, -1, -4, -7, -3, 1, -2, 4, 8, 5, -1, 3, 0, -3, -4, -7, 4, 3, 0, -1, -2, -5,
         -3, -6, -2, 4, 1, 5, -1, -4, 0, -3, 1, 5, 4, 8, 12, -1, 3, 7, -3, -4, 0, 4, 3, 7, -1, -2, 2,
         -3, -6, -7, 4, 1, 0, -1, -4, -5, -3, 1, 0, 4, 8, 7, -1, 3, 2, -3, -4, -5, 4, 3, 2, -1, -2, -3]


def dequant(quantvals, minval, delta, sequence):
    """_book_unquantize, maptype 1 (lib/sharedbook.c:243-268) from the encode side's integers"""
    out = []
    for e in range(27):
        last, div = 0.0, 1
        for k in range(3):
            val = QUANTLIST[(e // div) % quantvals] * delta + minval + last
            if sequence:
                last = val
            out.append(val)
            div *= quantvals
    return out


def check(lat):
    quantvals, minval, delta = lat
    assert (quantvals, minval, delta) == (3, -3, 1)
    assert dequant(quantvals, minval, delta, 0) == TEST4
    assert dequant(quantvals, minval, delta, 1) == TEST5


def test_oracle_on_the_references_selftest_books(oracle):
    out, fl = (C.c_int * 3)(), (C.c_float * 2)()
    oracle.lib.orc_book_lattice.argtypes = [C.c_long, C.c_long, C.c_long, C.c_long, C.POINTER(C.c_int), C.POINTER(C.c_float)]
    oracle.lib.orc_book_lattice(Q_MIN, Q_DELTA, 27, 3, out, fl)
    assert (fl[0], fl[1]) == (-3.0, 1.0)                       # _float32_unpack of the two header words
    check(tuple(out))
    # books 2 and 3 (:507-527): 3 entries of dimension 4 with the same packed floats
    oracle.lib.orc_book_lattice(Q_MIN, Q_DELTA, 3, 4, out, fl)
    assert (fl[0], fl[1]) == (-3.0, 1.0) and out[0] == 1       # _book_maptype1_quantvals(3, 4) = 1


def test_product_on_the_references_selftest_books():
    import vorbis_aotuv_lancer_amd as v
    out = (C.c_int * 3)()
    assert v.lib.vbm_host_book_lattice(Q_MIN, Q_DELTA, 27, 3, out) == 0
    check(tuple(out))
    assert v.lib.vbm_host_book_lattice(Q_MIN, Q_DELTA, 3, 4, out) == 0 and out[0] == 1
    # a few more lattice sizes against the definition: greatest v with v^dim <= entries (:174-207)
    for entries, dim in [(6561, 8), (625, 4), (81, 2), (289, 2), (3125, 5), (7, 3), (8, 3), (9, 3), (1, 5)]:
        assert v.lib.vbm_host_book_lattice(Q_MIN, Q_DELTA, entries, dim, out) == 0
        q = out[0]
        assert q ** dim <= entries < (q + 1) ** dim, (entries, dim, q)
