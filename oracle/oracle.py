"""ctypes front-end of the CPU oracle (oracle/klr_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (kmerlr_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libklr_oracle.so")
_lib = None


def build(force=False):
    """Compile the oracle with gcc (recipe: oracle/Makefile; make decides what is stale -- the driver library also
    depends on the product's klr_driver.cpp and headers).  Serialised by a file lock: several test ranks may start at once."""
    import fcntl
    os.makedirs(os.path.join(_HERE, "_build"), exist_ok=True)
    with open(os.path.join(_HERE, "_build", ".lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []) + ["all"])
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)
    return _LIB_PATH


class IterRec(C.Structure):
    _fields_ = [("iteration", C.c_int32), ("nonzero", C.c_int32), ("change", C.c_double),
                ("lambda_", C.c_double), ("loss", C.c_double), ("seconds", C.c_double)]


class _Csr(C.Structure):
    _fields_ = [("n", C.c_int64), ("dim", C.c_int64), ("rowptr", C.c_void_p),
                ("col", C.c_void_p), ("val", C.c_void_p)]


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp, vp, i64 = C.POINTER(C.c_double), C.c_void_p, C.c_int64
        L.klo_log_add.restype = C.c_double
        L.klo_log_add.argtypes = [C.c_double, C.c_double]
        L.klo_linear_pdf.restype = C.c_double
        L.klo_linear_pdf.argtypes = [C.POINTER(_Csr), i64, vp, vp, vp]
        L.klo_class_log_pdf.restype = C.c_double
        L.klo_class_log_pdf.argtypes = [C.POINTER(_Csr), i64, vp, vp, vp, C.c_int]
        L.klo_class_weights.argtypes = [vp, i64, vp]
        L.klo_gradient.restype = C.c_int
        L.klo_gradient.argtypes = [C.POINTER(_Csr), vp, vp, vp, C.c_double, vp, vp, vp]
        L.klo_gradient_mt.restype = C.c_int
        L.klo_gradient_mt.argtypes = [C.POINTER(_Csr), vp, vp, vp, vp, C.c_int]
        L.klo_loss.restype = C.c_double
        L.klo_loss.argtypes = [C.POINTER(_Csr), vp, vp, vp, C.c_double, vp, vp]
        L.klo_sort_abs_desc.argtypes = [vp, i64, vp]
        L.klo_compute_lambda.restype = C.c_double
        L.klo_compute_lambda.argtypes = [vp, i64, vp, i64]
        L.klo_select.restype = C.c_int
        L.klo_select.argtypes = [vp, i64, vp, i64, i64, C.c_double, C.c_double, dp, C.POINTER(i64)]
        L.klo_select_fixed.restype = C.c_int
        L.klo_select_fixed.argtypes = [vp, i64, vp, i64, C.c_double, i64, C.POINTER(i64)]
        L.klo_coeff_dim.restype = i64
        L.klo_coeff_dim.argtypes = [i64]
        L.klo_coeff_ind2sub.restype = i64
        L.klo_coeff_ind2sub.argtypes = [i64, i64, i64]
        L.klo_coeff_sub2ind.argtypes = [i64, i64, C.POINTER(i64), C.POINTER(i64)]
        L.klo_expand_cooccurrence.restype = C.c_int
        L.klo_expand_cooccurrence.argtypes = [C.POINTER(_Csr), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        L.klo_select_columns.restype = C.c_int
        L.klo_select_columns.argtypes = [C.POINTER(_Csr), vp, i64, vp, vp,
                                         C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
        L.klo_free.argtypes = [vp]
        L.klo_eval_stopping.restype = C.c_int
        L.klo_eval_stopping.argtypes = [vp, vp, i64, C.c_double, dp]
        L.klo_cd_solve.restype = C.c_int
        L.klo_cd_solve.argtypes = [C.POINTER(_Csr), vp, vp, vp, C.c_double, C.c_double, C.c_double,
                                   C.c_double, C.c_int, i64, vp, i64, C.POINTER(i64)]
        L.klo_transform_fit.restype = C.c_int
        L.klo_transform_fit.argtypes = [C.POINTER(_Csr), C.c_int, vp, vp, C.POINTER(C.c_int)]
        L.klo_prox_step.argtypes = [vp, vp, i64, C.c_double, C.c_double]
        L.klo_splitmix64.restype = C.c_uint64
        L.klo_splitmix64.argtypes = [C.c_uint64]
        L.klo_synth_motifs.argtypes = [C.c_uint64, i64, i64, vp]
        L.klo_synth_csr.argtypes = [C.c_uint64, i64, i64, i64, i64, i64, i64, vp, vp, vp, vp]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


class Csr:
    """Rows in the reference layout: int64 ascending indices + f64 values, index 0 = bias."""

    def __init__(self, n, dim, rowptr, col, val):
        self.n, self.dim = int(n), int(dim)
        self.rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        self.col = np.ascontiguousarray(col, dtype=np.int64)
        self.val = np.ascontiguousarray(val, dtype=np.float64)
        assert self.rowptr.shape == (self.n + 1,)
        self._c = _Csr(self.n, self.dim, _p(self.rowptr), _p(self.col), _p(self.val))

    @property
    def ref(self):
        return C.byref(self._c)

    @property
    def nnz(self):
        return int(self.rowptr[-1])

    @staticmethod
    def from_dense(x):
        """convert_scores with generate_features (scoresLr_data.go:104-117): every column is
        stored, zeros included; bias 1.0 prepended."""
        x = np.asarray(x, dtype=np.float64)
        n, m = x.shape
        rowptr = np.arange(n + 1, dtype=np.int64) * (m + 1)
        col = np.tile(np.arange(m + 1, dtype=np.int64), n)
        val = np.concatenate([np.ones((n, 1)), x], axis=1).reshape(-1)
        return Csr(n, m + 1, rowptr, col, val)

    def to_dense(self):
        d = np.zeros((self.n, self.dim))
        for i in range(self.n):
            s, e = self.rowptr[i], self.rowptr[i + 1]
            d[i, self.col[s:e]] = self.val[s:e]
        return d


def log_add(a, b):
    return lib().klo_log_add(a, b)


def linear_pdf(x, i, theta, scale=None, offset=None):
    theta, scale, offset = _f64(theta), _f64(scale), _f64(offset)
    return lib().klo_linear_pdf(x.ref, i, _p(theta), _p(scale), _p(offset))


def class_log_pdf(x, i, theta, y, scale=None, offset=None):
    theta, scale, offset = _f64(theta), _f64(scale), _f64(offset)
    return lib().klo_class_log_pdf(x.ref, i, _p(theta), _p(scale), _p(offset), int(y))


def class_weights(labels):
    labels = np.ascontiguousarray(labels, dtype=np.uint8)
    cw = np.zeros(2)
    lib().klo_class_weights(_p(labels), len(labels), _p(cw))
    return cw


def gradient(x, labels, theta, cw=(1.0, 1.0), lam=0.0, scale=None, offset=None, threads=1):
    """threads > 1: klo_gradient_mt (rows over OpenMP threads; bench.py's all-core figure), no transform / lambda."""
    labels = np.ascontiguousarray(labels, dtype=np.uint8)
    theta, scale, offset, cw = _f64(theta), _f64(scale), _f64(offset), _f64(cw)
    assert theta.shape == (x.dim,)
    g = np.zeros(x.dim)
    if threads > 1:
        assert scale is None and offset is None and lam == 0.0
        rc = lib().klo_gradient_mt(x.ref, _p(labels), _p(theta), _p(cw), _p(g), int(threads))
        assert rc == 0
        return g
    lib().klo_gradient(x.ref, _p(labels), _p(theta), _p(cw), lam, _p(scale), _p(offset), _p(g))
    return g


def loss(x, labels, theta, cw=(1.0, 1.0), lam=0.0, scale=None, offset=None):
    labels = np.ascontiguousarray(labels, dtype=np.uint8)
    theta, scale, offset, cw = _f64(theta), _f64(scale), _f64(offset), _f64(cw)
    assert theta.shape == (x.dim,)
    return lib().klo_loss(x.ref, _p(labels), _p(theta), _p(cw), lam, _p(scale), _p(offset))


def sort_abs_desc(g):
    g = _f64(g)
    order = np.zeros(len(g), dtype=np.int64)
    lib().klo_sort_abs_desc(_p(g), len(g), _p(order))
    return order


def select(g, b, c_prev, N, lambda_prev=0.0, eps_lambda=0.0):
    """g = gradient[1:], b = uint8 mask of length m+1 (modified copy is returned).
    Returns (ok, lambda, c, b)."""
    g = _f64(g)
    b = np.array(b, dtype=np.uint8, copy=True)
    assert b.shape == (len(g) + 1,)
    lam, c = C.c_double(0.0), C.c_int64(0)
    ok = lib().klo_select(_p(g), len(g), _p(b), c_prev, N, lambda_prev, eps_lambda,
                          C.byref(lam), C.byref(c))
    return bool(ok), lam.value, c.value, b


def select_fixed(g_full, b, c_prev, lam, max_features=0):
    g_full = _f64(g_full)
    b = np.array(b, dtype=np.uint8, copy=True)
    c = C.c_int64(0)
    ok = lib().klo_select_fixed(_p(g_full), len(g_full) - 1, _p(b), c_prev, lam, max_features, C.byref(c))
    return bool(ok), c.value, b


def select_columns(x, cols, scale=None, offset=None):
    cols = np.ascontiguousarray(cols, dtype=np.int64)
    scale, offset = _f64(scale), _f64(offset)
    rp, ci, cv = C.c_void_p(), C.c_void_p(), C.c_void_p()
    lib().klo_select_columns(x.ref, _p(cols), len(cols), _p(scale), _p(offset),
                             C.byref(rp), C.byref(ci), C.byref(cv))
    n = x.n
    rowptr = np.ctypeslib.as_array(C.cast(rp, C.POINTER(C.c_int64)), shape=(n + 1,)).copy()
    nnz = int(rowptr[-1])
    col = np.ctypeslib.as_array(C.cast(ci, C.POINTER(C.c_int64)), shape=(max(nnz, 1),))[:nnz].copy()
    val = np.ctypeslib.as_array(C.cast(cv, C.POINTER(C.c_double)), shape=(max(nnz, 1),))[:nnz].copy()
    for p in (rp, ci, cv):
        lib().klo_free(p)
    return Csr(n, len(cols), rowptr, col, val)


def coeff_dim(n):
    return int(lib().klo_coeff_dim(int(n)))


def coeff_ind2sub(n, k1, k2):
    return int(lib().klo_coeff_ind2sub(int(n), int(k1), int(k2)))


def coeff_sub2ind(n, i):
    a, b = C.c_int64(), C.c_int64()
    lib().klo_coeff_sub2ind(int(n), int(i), C.byref(a), C.byref(b))
    return a.value, b.value


def expand_cooccurrence(x):
    """Co-occurrence feature matrix of x in CoeffIndex column order (singles, then pairs)."""
    rp, ci, cv = C.c_void_p(), C.c_void_p(), C.c_void_p()
    lib().klo_expand_cooccurrence(x.ref, C.byref(rp), C.byref(ci), C.byref(cv))
    n = x.n
    rowptr = np.ctypeslib.as_array(C.cast(rp, C.POINTER(C.c_int64)), shape=(n + 1,)).copy()
    nnz = int(rowptr[-1])
    col = np.ctypeslib.as_array(C.cast(ci, C.POINTER(C.c_int64)), shape=(max(nnz, 1),))[:nnz].copy()
    val = np.ctypeslib.as_array(C.cast(cv, C.POINTER(C.c_double)), shape=(max(nnz, 1),))[:nnz].copy()
    for p in (rp, ci, cv):
        lib().klo_free(p)
    return Csr(n, coeff_dim(x.dim - 1), rowptr, col, val)


def eval_stopping(xs, x1, eps):
    xs, x1 = _f64(xs), _f64(x1)
    d = C.c_double(0.0)
    stop = lib().klo_eval_stopping(_p(xs), _p(x1), len(xs), eps, C.byref(d))
    return bool(stop), d.value


def cd_solve(x, labels, theta, cw=(1.0, 1.0), l1reg=0.0, l2reg=0.0, eps=0.0, eps_loss=1e-8,
             eval_loss=True, max_iter=0, recs_cap=100000):
    """Returns (theta, records[list of dict])."""
    labels = np.ascontiguousarray(labels, dtype=np.uint8)
    theta = np.array(theta, dtype=np.float64, copy=True)
    cw = _f64(cw)
    assert theta.shape == (x.dim,)
    recs = (IterRec * recs_cap)()
    n_recs = C.c_int64(0)
    lib().klo_cd_solve(x.ref, _p(labels), _p(theta), _p(cw), l1reg, l2reg, eps, eps_loss,
                       int(eval_loss), max_iter, C.cast(recs, C.c_void_p), recs_cap, C.byref(n_recs))
    out = [dict(iteration=r.iteration, nonzero=r.nonzero, change=r.change, lambda_=r.lambda_,
                loss=r.loss, seconds=r.seconds) for r in recs[:min(n_recs.value, recs_cap)]]
    return theta, out


TRANSFORMS = {"standardizer": 1, "variance-scaler": 2, "max-abs-scaler": 3, "mean-scaler": 4}


def transform_fit(x, kind):
    """Returns (scale, offset or None)."""
    scale, offset = np.zeros(x.dim), np.zeros(x.dim)
    ho = C.c_int(0)
    rc = lib().klo_transform_fit(x.ref, TRANSFORMS[kind], _p(scale), _p(offset), C.byref(ho))
    assert rc == 0
    return scale, (offset if ho.value else None)


def prox_step(theta, g, step, l1reg):
    theta = np.array(theta, dtype=np.float64, copy=True)
    g = _f64(g)
    lib().klo_prox_step(_p(theta), _p(g), len(theta), step, l1reg)
    return theta


def splitmix64(x):
    return lib().klo_splitmix64(C.c_uint64(x & 0xFFFFFFFFFFFFFFFF))


def synth_motifs(seed, m, S):
    ids = np.zeros(max(S, 1), dtype=np.int64)
    lib().klo_synth_motifs(C.c_uint64(seed), m, S, _p(ids))
    return ids[:S]


def synth_csr(seed, row0, n_local, n_total, m, draws, S):
    """Returns (Csr, labels uint8)."""
    rowptr = np.zeros(n_local + 1, dtype=np.int64)
    col = np.zeros(n_local * (draws + 1), dtype=np.int64)
    val = np.zeros(n_local * (draws + 1), dtype=np.float64)
    labels = np.zeros(n_local, dtype=np.uint8)
    lib().klo_synth_csr(C.c_uint64(seed), row0, n_local, n_total, m, draws, S,
                        _p(rowptr), _p(col), _p(val), _p(labels))
    nnz = int(rowptr[-1])
    return Csr(n_local, m + 1, rowptr, col[:nnz].copy(), val[:nnz].copy()), labels


def read_scores_table(path, header=False):
    """read_scores_table, scoresLr_data.go:56-100: comma separated, fields trimmed."""
    names, rows = [], []
    with open(path) as f:
        lines = [l.rstrip("\n") for l in f]
    it = iter(lines)
    if header:
        for l in it:
            if len(l) == 0:
                continue
            names = [s for s in l.split(",") if s != ""]
            break
    for l in it:
        if len(l) == 0:
            continue
        rows.append([float(s.strip()) for s in l.split(",") if s != ""])
    return names, np.array(rows, dtype=np.float64)
