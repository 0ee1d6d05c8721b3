"""ctypes binding of include/kmerlr_driver.h: the leapfrog path driver (KmerLrEstimator.Estimate).

`Estimator` works with any library that exports the klr_estimator_* entry points and a klr_backend
table; the product passes libkmerlr_hip.so + klr_hip_backend (see `hip_estimator`)."""
import ctypes as C

import numpy as np

from . import _capi


class Backend(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ctx", "last_error", "mat_shape", "mat_destroy", "class_weights", "gradient",
                                          "loss", "select_columns", "sweep_select", "cd_solve", "set_active_rows",
                                          "log_pdf", "expand_cooccurrence", "transform_fit", "ctx_fork", "ctx_release",
                                          "mat_view", "solve_run_begin", "solve_run_end")]


class Config(C.Structure):
    """klr_config: hot-path fields of the reference's Config (kmerLr.go:32-62)."""
    _fields_ = [("lambda_", C.c_double), ("max_features", C.c_int64), ("max_epochs", C.c_int64),
                ("max_iterations", C.c_int64), ("epsilon", C.c_double), ("epsilon_lambda", C.c_double),
                ("epsilon_loss", C.c_double), ("eval_loss", C.c_int), ("balance", C.c_int), ("penalty_free", C.c_int),
                ("save_trace", C.c_int), ("save_path", C.c_int), ("verbose", C.c_int), ("l2reg", C.c_double),
                ("cooccurrence", C.c_int), ("copreselection", C.c_int64), ("validation_size", C.c_double),
                ("ensemble_size", C.c_int), ("ensemble_summary", C.c_int), ("data_transform", C.c_int), ("threads_cv", C.c_int), ("threads", C.c_int)]


def bind_driver(L):
    vp, i64 = C.c_void_p, C.c_int64
    L.klr_config_default.argtypes = [C.POINTER(Config)]
    L.klr_estimator_create.restype = C.c_int
    L.klr_estimator_create.argtypes = [C.POINTER(Backend), C.POINTER(Config), C.POINTER(vp)]
    L.klr_estimator_destroy.argtypes = [vp]
    L.klr_estimator_error.restype = C.c_char_p
    L.klr_estimator_error.argtypes = [vp]
    L.klr_estimator_set_model.restype = C.c_int
    L.klr_estimator_set_model.argtypes = [vp, vp, vp, i64]
    L.klr_estimator_estimate.restype = C.c_int
    L.klr_estimator_estimate.argtypes = [vp, vp, vp, i64, vp, vp]
    L.klr_estimator_n_classifiers.restype = i64
    L.klr_estimator_n_classifiers.argtypes = [vp]
    L.klr_estimator_classifier.restype = C.c_int
    L.klr_estimator_classifier.argtypes = [vp, i64, C.POINTER(i64), C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_double),
                                           C.POINTER(vp), C.POINTER(vp)]
    L.klr_estimator_trace.restype = C.c_int
    L.klr_estimator_trace.argtypes = [vp, C.POINTER(vp), C.POINTER(i64)]
    L.klr_estimator_epochs.restype = C.c_int
    L.klr_estimator_epochs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(i64)]
    for name in ("klr_estimator_save_trace", "klr_estimator_save_path"):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [vp, C.c_char_p]
    L.klr_estimator_crossvalidate.restype = C.c_int
    L.klr_estimator_crossvalidate.argtypes = [vp, vp, i64, i64, C.c_int, i64, vp, i64, vp, vp, vp, vp, vp, vp]
    L.klr_cv_groups.argtypes = [i64, C.c_int, i64, i64, i64, vp]
    L.klr_cv_groups_val.argtypes = [i64, C.c_int, C.c_double, i64, i64, i64, vp, vp]
    L.klr_estimator_base_cols.restype = C.c_int
    L.klr_estimator_base_cols.argtypes = [vp, C.POINTER(vp), C.POINTER(i64)]
    L.klr_estimator_fold.restype = vp
    L.klr_estimator_fold.argtypes = [vp, C.c_int]
    L.klr_save_crossvalidation.restype = C.c_int
    L.klr_save_crossvalidation.argtypes = [C.c_char_p, vp, vp, vp, i64, C.c_int]
    L.klr_save_crossvalidation_loss.restype = C.c_int
    L.klr_save_crossvalidation_loss.argtypes = [C.c_char_p, vp, vp, C.c_int]
    L.klr_estimator_save_model.restype = C.c_int
    L.klr_estimator_save_model.argtypes = [vp, i64, C.c_char_p, C.c_char_p, vp]
    L.klr_estimator_save_model_pairs.restype = C.c_int
    L.klr_estimator_save_model_pairs.argtypes = [vp, i64, C.c_char_p, C.c_char_p, vp, vp]
    L.klr_estimator_estimate_ensemble.restype = C.c_int
    L.klr_estimator_estimate_ensemble.argtypes = [vp, vp, i64, i64, C.c_int, i64, vp, i64, vp, vp]
    L.klr_estimator_save_ensemble.restype = C.c_int
    L.klr_estimator_save_ensemble.argtypes = [vp, i64, C.c_int, i64, C.c_char_p, C.c_char_p, vp, C.c_char_p]
    L.klr_estimator_save_ensemble_pairs.restype = C.c_int
    L.klr_estimator_save_ensemble_pairs.argtypes = [vp, i64, C.c_int, i64, C.c_char_p, C.c_char_p, vp, C.c_char_p, vp, i64]
    L.klr_cooccurrence_prepare.restype = C.c_int
    L.klr_cooccurrence_prepare.argtypes = [C.POINTER(Backend), C.POINTER(Config), vp, vp, vp, C.POINTER(vp), vp, C.POINTER(i64),
                                           C.c_char_p, i64]
    return L


def cooccurrence_prepare(L, backend, cfg, mat_handle, dim, pre_scale=None, pre_offset=None):
    """klr_cooccurrence_prepare: returns (expanded matrix handle, base_cols).  The caller destroys the handle
    through the backend's mat_destroy."""
    out = C.c_void_p()
    cols = np.zeros(dim, dtype=np.int64)
    nb1 = C.c_int64()
    err = C.create_string_buffer(512)
    ps = None if pre_scale is None else np.ascontiguousarray(pre_scale, dtype=np.float64)
    po = None if pre_offset is None else np.ascontiguousarray(pre_offset, dtype=np.float64)
    rc = L.klr_cooccurrence_prepare(C.byref(backend), C.byref(cfg), mat_handle, None if ps is None else ps.ctypes.data,
                                    None if po is None else po.ctypes.data, C.byref(out), cols.ctypes.data, C.byref(nb1), err, 512)
    if rc != 0:
        raise _capi.KlrError("klr_cooccurrence_prepare failed (%d): %s" % (rc, err.value.decode()))
    return out, cols[:nb1.value].copy()


def default_config(L, **kw):
    cfg = Config()
    L.klr_config_default(C.byref(cfg))
    for k, v in kw.items():
        key = "lambda_" if k == "lambda_" or k == "lam" else k
        if not hasattr(cfg, key):
            raise TypeError("unknown config field %s" % k)
        setattr(cfg, key, v)
    return cfg


class Estimator:
    """KmerLrEstimator / ScoresLrEstimator: leapfrog path driver over a backend table."""

    def __init__(self, L, backend, **config):
        self.L = L
        self.backend = backend
        self.cfg = default_config(L, **config)
        self._h = C.c_void_p()
        rc = L.klr_estimator_create(C.byref(backend), C.byref(self.cfg), C.byref(self._h))
        if rc != 0:
            raise _capi.KlrError("klr_estimator_create failed (%d)" % rc)

    def close(self):
        if self._h:
            self.L.klr_estimator_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise _capi.KlrError("driver error %d: %s" % (rc, (self.L.klr_estimator_error(self._h) or b"").decode()))

    def set_model(self, cols, theta):
        cols = np.ascontiguousarray(cols, dtype=np.int64)
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        self._check(self.L.klr_estimator_set_model(self._h, cols.ctypes.data, theta.ctypes.data, len(cols)))

    def estimate(self, mat_handle, lambda_auto=(), scale=None, offset=None):
        la = np.ascontiguousarray(list(lambda_auto), dtype=np.int64)
        scale = None if scale is None else np.ascontiguousarray(scale, dtype=np.float64)
        offset = None if offset is None else np.ascontiguousarray(offset, dtype=np.float64)
        self._check(self.L.klr_estimator_estimate(self._h, mat_handle, la.ctypes.data if len(la) else None, len(la),
                                                  None if scale is None else scale.ctypes.data,
                                                  None if offset is None else offset.ctypes.data))
        return self.classifiers()

    def fold(self, f):
        """Estimator of fold f of the last cross-validation (owned by this one)."""
        h = self.L.klr_estimator_fold(self._h, f)
        if not h:
            return None
        fe = Estimator.__new__(Estimator)
        fe.L, fe.backend, fe.cfg, fe._h = self.L, self.backend, self.cfg, C.c_void_p(h)
        fe.close = lambda: None
        return fe

    def base_cols(self):
        """Base columns of a fold's pair matrix (co-preselection inside the fold)."""
        p, n = C.c_void_p(), C.c_int64()
        self._check(self.L.klr_estimator_base_cols(self._h, C.byref(p), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, dtype=np.int64)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int64)), shape=(n.value,)).copy()

    def crossvalidate(self, mat_handle, n_local, k_fold, lambda_auto, seed=1, row0=0, n_all=0, scale=None, offset=None):
        """Returns dict(predictions[nla, n_local], groups[n_local], loss_train[nla, k], loss_test[nla, k])."""
        la = np.ascontiguousarray(list(lambda_auto), dtype=np.int64)
        nla = len(la)
        n_all = n_all or n_local
        pred = np.zeros((nla, n_local)); groups = np.zeros(n_local, dtype=np.int32)
        lt = np.zeros((nla, k_fold)); ls = np.zeros((nla, k_fold))
        scale = None if scale is None else np.ascontiguousarray(scale, dtype=np.float64)
        offset = None if offset is None else np.ascontiguousarray(offset, dtype=np.float64)
        self._check(self.L.klr_estimator_crossvalidate(
            self._h, mat_handle, row0, n_all, k_fold, seed, la.ctypes.data, nla,
            None if scale is None else scale.ctypes.data, None if offset is None else offset.ctypes.data,
            pred.ctypes.data, groups.ctypes.data, lt.ctypes.data, ls.ctypes.data))
        return dict(predictions=pred, groups=groups, loss_train=lt, loss_test=ls, classifiers=self.classifiers())

    def estimate_ensemble(self, mat_handle, n_local, ensemble_size, lambda_auto, seed=1, row0=0, n_all=0, scale=None, offset=None):
        """Member-major list of classifiers (member k, lambda_auto j at k*len(lambda_auto) + j)."""
        la = np.ascontiguousarray(list(lambda_auto), dtype=np.int64)
        scale = None if scale is None else np.ascontiguousarray(scale, dtype=np.float64)
        offset = None if offset is None else np.ascontiguousarray(offset, dtype=np.float64)
        self._check(self.L.klr_estimator_estimate_ensemble(
            self._h, mat_handle, row0, n_all or n_local, ensemble_size, seed, la.ctypes.data, len(la),
            None if scale is None else scale.ctypes.data, None if offset is None else offset.ctypes.data))
        return self.classifiers()

    def save_ensemble(self, j, ensemble_size, n_lambda_auto, filename, type_name="scoresLr", names=None, summary="mean"):
        arr = None
        if names is not None:
            arr = (C.c_char_p * len(names))(*[s.encode() for s in names])
        self._check(self.L.klr_estimator_save_ensemble(self._h, j, ensemble_size, n_lambda_auto, filename.encode(), type_name.encode(),
                                                       None if arr is None else C.cast(arr, C.c_void_p), summary.encode()))

    def save_ensemble_pairs(self, j, ensemble_size, n_lambda_auto, filename, base_cols, type_name="scoresLr", names=None, summary="mean"):
        """An ensemble of co-occurrence models: base_cols as cooccurrence_prepare / base_cols() report them."""
        arr = None
        if names is not None:
            arr = (C.c_char_p * len(names))(*[s.encode() for s in names])
        bc = np.ascontiguousarray(base_cols, dtype=np.int64)
        self._check(self.L.klr_estimator_save_ensemble_pairs(self._h, j, ensemble_size, n_lambda_auto, filename.encode(), type_name.encode(),
                                                             None if arr is None else C.cast(arr, C.c_void_p), summary.encode(),
                                                             bc.ctypes.data, len(bc)))

    def classifiers(self):
        out = []
        for i in range(self.L.klr_estimator_n_classifiers(self._h)):
            c1, lam = C.c_int64(), C.c_double()
            cols, th, sc, of = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
            self._check(self.L.klr_estimator_classifier(self._h, i, C.byref(c1), C.byref(cols), C.byref(th), C.byref(lam),
                                                        C.byref(sc), C.byref(of)))
            n = c1.value

            def arr(p, t):
                if not p:
                    return None
                return np.ctypeslib.as_array(C.cast(p, C.POINTER(t)), shape=(n,)).copy()
            out.append(dict(cols=arr(cols.value, C.c_int64), theta=arr(th.value, C.c_double), lam=lam.value,
                            scale=arr(sc.value, C.c_double), offset=arr(of.value, C.c_double)))
        return out

    def trace(self):
        p, n = C.c_void_p(), C.c_int64()
        self._check(self.L.klr_estimator_trace(self._h, C.byref(p), C.byref(n)))
        if n.value == 0:
            return []
        recs = C.cast(p, C.POINTER(_capi.IterRec))
        return [dict(iteration=recs[i].iteration, nonzero=recs[i].nonzero, change=recs[i].change,
                     lambda_=recs[i].lambda_, loss=recs[i].loss, seconds=recs[i].seconds) for i in range(n.value)]

    def epochs(self):
        pl, pn, n = C.c_void_p(), C.c_void_p(), C.c_int64()
        self._check(self.L.klr_estimator_epochs(self._h, C.byref(pl), C.byref(pn), C.byref(n)))
        if n.value == 0:
            return [], []
        lam = np.ctypeslib.as_array(C.cast(pl, C.POINTER(C.c_double)), shape=(n.value,)).copy()
        nz = np.ctypeslib.as_array(C.cast(pn, C.POINTER(C.c_int64)), shape=(n.value,)).copy()
        return list(lam), list(nz)

    def save_trace(self, filename):
        self._check(self.L.klr_estimator_save_trace(self._h, filename.encode()))

    def save_path(self, filename):
        self._check(self.L.klr_estimator_save_path(self._h, filename.encode()))

    def save_model(self, i, filename, type_name="scoresLr", names=None):
        arr = None
        if names is not None:
            arr = (C.c_char_p * len(names))(*[s.encode() for s in names])
        self._check(self.L.klr_estimator_save_model(self._h, i, filename.encode(), type_name.encode(),
                                                    None if arr is None else C.cast(arr, C.c_void_p)))


    def save_model_pairs(self, i, filename, pairs, type_name="scoresLr", names=None):
        """pairs: [(f1, f2)] 0-based original features of every selected column (co-occurrence models)."""
        arr = None
        if names is not None:
            arr = (C.c_char_p * len(names))(*[s.encode() for s in names])
        pr = np.ascontiguousarray(np.asarray(pairs, dtype=np.int64).reshape(-1))
        self._check(self.L.klr_estimator_save_model_pairs(self._h, i, filename.encode(), type_name.encode(),
                                                          None if arr is None else C.cast(arr, C.c_void_p), pr.ctypes.data))


def hip_estimator(ctx, **config):
    """Estimator over the HIP backend of a kmerlr_amd.Context."""
    L = bind_driver(_capi.lib())
    L.klr_hip_backend.restype = C.c_int
    L.klr_hip_backend.argtypes = [C.c_void_p, C.POINTER(Backend)]
    be = Backend()
    rc = L.klr_hip_backend(ctx._h, C.byref(be))
    if rc != 0:
        raise _capi.KlrError("klr_hip_backend failed (%d)" % rc)
    return Estimator(L, be, **config)
