"""Test helper: the product's host driver (klr_driver.cpp) linked against the oracle-served backend
table (oracle/_build/libklr_oracle_driver.so).  CPU only; optional row sharding through a hook."""
import ctypes as C
import os

import numpy as np

from kmerlr_amd import _driver as D
from oracle import oracle as O

_LIB = None
AR_FN = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_int64, C.c_void_p)


def lib():
    global _LIB
    if _LIB is None:
        O.build()
        path = os.path.join(os.path.dirname(O._LIB_PATH), "libklr_oracle_driver.so")
        L = C.CDLL(path)
        D.bind_driver(L)
        L.klo_backend_table.argtypes = [C.POINTER(D.Backend)]
        L.klo_backend_mat.restype = C.c_void_p
        L.klo_backend_mat.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.klo_backend_mat_destroy.argtypes = [C.c_void_p]
        L.klo_set_sharding.argtypes = [C.c_int64, AR_FN, C.c_void_p]
        _LIB = L
    return _LIB


def estimator(**config):
    L = lib()
    be = D.Backend()
    L.klo_backend_table(C.byref(be))
    return D.Estimator(L, be, **config)


def matrix(x, labels, n_total=0):
    L = lib()
    labels = np.ascontiguousarray(labels, dtype=np.uint8)
    return L.klo_backend_mat(x.n, x.dim, x.rowptr.ctypes.data, x.col.ctypes.data, x.val.ctypes.data, labels.ctypes.data,
                             n_total)


def free_matrix(h):
    lib().klo_backend_mat_destroy(h)


_keep = []


def set_sharding(n_total, fn):
    """fn(np_array) sums in place across ranks; None switches sharding off."""
    L = lib()
    if fn is None:
        L.klo_set_sharding(0, C.cast(None, AR_FN), None)
        return

    def tramp(buf, count, user):
        a = np.ctypeslib.as_array(buf, shape=(count,))
        fn(a)
    cb = AR_FN(tramp)
    _keep.append(cb)
    L.klo_set_sharding(n_total, cb, None)
