"""ctypes binding of include/kmerlr.h (libkmerlr_hip.so).  There is no fallback: if the library is
missing or no gfx950 device is present, every entry point raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_lib", "libkmerlr_hip.so")
_lib = None


class KlrError(RuntimeError):
    pass


class IterRec(C.Structure):
    _fields_ = [("iteration", C.c_int32), ("nonzero", C.c_int32), ("change", C.c_double),
                ("lambda_", C.c_double), ("loss", C.c_double), ("seconds", C.c_double)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p)
REDUCE_SCATTER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)

# name -> (restype, argtypes); also the export list checked by tests/test_abi.py
_vp, _i64, _dbl, _int = C.c_void_p, C.c_int64, C.c_double, C.c_int
_PVP = C.POINTER(C.c_void_p)
SIGNATURES = {
    "klr_ctx_create": (_int, [_int, _PVP]),
    "klr_ctx_destroy": (None, [_vp]),
    "klr_solve_run_begin": (None, [_vp]),
    "klr_solve_run_end": (None, [_vp]),
    "klr_ctx_fork": (_int, [_vp, C.POINTER(_vp)]),
    "klr_ctx_release": (None, [_vp]),
    "klr_last_error": (C.c_char_p, [_vp]),
    "klr_ctx_set_stream": (_int, [_vp, _vp]),
    "klr_ctx_synchronize": (_int, [_vp]),
    "klr_ctx_set_allreduce": (_int, [_vp, ALLREDUCE_FN, _vp, _int, _int]),
    "klr_ctx_set_collectives": (_int, [_vp, REDUCE_SCATTER_FN, ALLGATHER_FN, _vp]),
    "klr_rccl_unique_id": (_int, [_vp]),
    "klr_ctx_init_rccl": (_int, [_vp, _vp, _int, _int]),
    "klr_ctx_comm_info": (_int, [_vp, C.POINTER(_int), C.POINTER(_int), C.POINTER(_int)]),
    "klr_allreduce_time": (_int, [_vp, _i64, _int, _int, C.POINTER(_dbl)]),
    "klr_shm_region_bytes": (_i64, [_int]),
    "klr_shm_region_init": (_int, [_vp, _i64, _int]),
    "klr_shm_region_abort": (None, [_vp]),
    "klr_ctx_init_shm": (_int, [_vp, _vp, _i64, _int, _int]),
    "klr_allreduce_host": (_int, [_vp, _vp, _i64, _int]),
    "klr_ctx_comm_selfcheck": (_int, [_vp]),
    "klr_ctx_comm_stats": (_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_int)]),
    "klr_ctx_comm_ranged": (_int, [_vp, C.POINTER(_i64)]),
    "klr_ctx_comm_probe": (_int, [_vp, C.POINTER(_i64), C.POINTER(_dbl), C.POINTER(_dbl), C.POINTER(_dbl), C.POINTER(_int), C.POINTER(_int)]),
    "klr_ctx_set_option": (_int, [_vp, C.c_char_p, C.c_char_p]),
    "klr_ctx_reload_env": (_int, [_vp]),
    "klr_mat_from_csr": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _PVP]),
    "klr_mat_from_sequences": (_int, [_vp, _vp, _vp, _i64, _int, _int, _int, _int, _vp, _i64, _PVP]),
    "klr_kmer_classes_create": (_int, [_vp, _PVP, _vp, _i64]),
    "klr_kmer_classes_destroy": (None, [_vp]),
    "klr_kmer_classes_count": (_i64, [_vp]),
    "klr_kmer_class_name": (_int, [_vp, _i64, _vp, _i64]),
    "klr_kmer_class_find": (_i64, [_vp, C.c_char_p]),
    "klr_mat_from_sequences_classes": (_int, [_vp, _vp, _vp, _i64, _vp, _int, _vp, _i64, _PVP]),
    "klr_mat_destroy": (None, [_vp]),
    "klr_mat_shape": (_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "klr_mat_layout": (_int, [_vp, C.POINTER(_int), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64),
                              C.POINTER(_int), C.POINTER(_int)]),
    "klr_mat_to_csr": (_int, [_vp, _vp, _vp, _vp, _vp]),
    "klr_mat_set_active_rows": (_int, [_vp, _vp, _vp, _i64]),
    "klr_mat_view": (_int, [_vp, _vp, C.POINTER(_vp)]),
    "klr_mat_expand_cooccurrence": (_int, [_vp, _vp, _PVP]),
    "klr_coeff_dim": (_i64, [_i64]),
    "klr_coeff_ind2sub": (_i64, [_i64, _i64, _i64]),
    "klr_coeff_sub2ind": (None, [_i64, _i64, _vp, _vp]),
    "klr_class_weights": (_int, [_vp, _vp, _vp]),
    "klr_gradient": (_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "klr_loss": (_int, [_vp, _vp, _vp, _vp, _dbl, C.POINTER(_dbl)]),
    "klr_log_pdf": (_int, [_vp, _vp, _vp, _vp]),
    "klr_transform_fit": (_int, [_vp, _vp, _int, _vp, _vp, C.POINTER(_int)]),
    "klr_select": (_int, [_vp, _vp, _i64, _vp, _i64, _i64, _dbl, _dbl, C.POINTER(_dbl), C.POINTER(_i64),
                          C.POINTER(_int)]),
    "klr_select_fixed": (_int, [_vp, _vp, _i64, _vp, _i64, _dbl, _i64, C.POINTER(_i64), C.POINTER(_int)]),
    "klr_mat_select_columns": (_int, [_vp, _vp, _vp, _i64, _vp, _vp, _PVP]),
    "klr_sweep_select": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _dbl, _dbl, _vp, _vp, _vp, C.POINTER(_dbl),
                                C.POINTER(_i64), C.POINTER(_int), C.POINTER(_dbl), _vp]),
    "klr_cd_solve": (_int, [_vp, _vp, _vp, _vp, _dbl, _dbl, _dbl, _dbl, _int, _i64, _vp, _i64, C.POINTER(_i64)]),
    "klr_synth_csr": (_int, [C.c_uint64, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _int]),
    "klr_prof_enable": (_int, [_vp, _int]),
    "klr_prof_reserve": (_int, [_vp, _i64]),
    "klr_prof_read": (_int, [_vp, C.POINTER(_i64), C.POINTER(_dbl), _int]),
    "klr_prof_read_cd": (_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_dbl), _int]),
    "klr_prof_evict": (_int, [_vp, _i64]),
    "klr_mat_stream_info": (_int, [_vp, C.POINTER(_int), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "klr_version": (C.c_char_p, []),
}


def lib():
    """Loads libkmerlr_hip.so; raises if it has not been built (python -m kmerlr_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise KlrError("libkmerlr_hip.so is missing (%s): build it with `python -m kmerlr_amd.build`; "
                           "there is no CPU fallback" % LIB_PATH)
        try:
            import torch  # noqa: F401  (one HIP runtime per process: let torch's copy load first)
        except Exception:  # pragma: no cover
            pass
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def _check(ctx_handle, rc):
    if rc != 0:
        msg = lib().klr_last_error(ctx_handle)
        raise KlrError("kmerlr error %d: %s" % (rc, (msg or b"").decode()))


class Context:
    """klr_ctx: one GPU, one stream, single-threaded."""

    _live = None

    def __init__(self, device_id=0):
        self._h = C.c_void_p()
        rc = lib().klr_ctx_create(device_id, C.byref(self._h))
        if rc != 0:
            raise KlrError("klr_ctx_create failed (%d): %s" % (rc, (lib().klr_last_error(None) or b"").decode()))
        self._cb = None
        if Context._live is None:
            import weakref
            Context._live = weakref.WeakSet()
        Context._live.add(self)

    @staticmethod
    def live():
        """Contexts that have not been closed (tests set KLR_* switches on all of them)."""
        return [c for c in (Context._live or ()) if c._h]

    def fork(self):
        """klr_ctx_fork: a second context on this device with this one's switches (own stream and workspaces), for another
        host thread; release() hands it back to this context."""
        c = Context.__new__(Context)
        c._h = C.c_void_p()
        self.check(lib().klr_ctx_fork(self._h, C.byref(c._h)))
        c._cb = None
        c._forked = True
        Context._live.add(c)
        return c

    def release(self):
        if self._h:
            lib().klr_ctx_release(self._h)
            self._h = C.c_void_p()

    def close(self):
        if self._h:
            if getattr(self, "_forked", False):
                lib().klr_ctx_release(self._h)
            else:
                lib().klr_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        _check(self._h, rc)

    def synchronize(self):
        self.check(lib().klr_ctx_synchronize(self._h))

    def set_stream(self, stream_ptr):
        self.check(lib().klr_ctx_set_stream(self._h, C.c_void_p(stream_ptr)))

    def set_allreduce(self, fn, rank, nranks):
        """fn(dev_ptr:int, count:int, dtype:int (0 f64, 1 int64), stream:int) -> None, sums in place."""
        def tramp(user, buf, count, dtype, stream):
            try:
                fn(buf, count, dtype, stream)
                return 0
            except Exception as e:  # pragma: no cover
                print("allreduce hook failed:", e)
                return 1
        self._cb = ALLREDUCE_FN(tramp)
        self.check(lib().klr_ctx_set_allreduce(self._h, self._cb, None, rank, nranks))

    def set_collectives(self, rs_fn, ag_fn):
        """Reduce-scatter + all-gather hooks (klr_ctx_set_collectives), after set_allreduce:
        rs_fn(send_ptr, recv_ptr, recv_count, dtype, stream), ag_fn(send_ptr, recv_ptr, send_bytes, stream)."""
        def t_rs(user, send, recv, count, dtype, stream):
            try:
                rs_fn(send, recv, count, dtype, stream)
                return 0
            except Exception as e:  # pragma: no cover
                print("reduce-scatter hook failed:", e)
                return 1

        def t_ag(user, send, recv, nbytes, stream):
            try:
                ag_fn(send, recv, nbytes, stream)
                return 0
            except Exception as e:  # pragma: no cover
                print("all-gather hook failed:", e)
                return 1
        self._cb_rs, self._cb_ag = REDUCE_SCATTER_FN(t_rs), ALLGATHER_FN(t_ag)
        self.check(lib().klr_ctx_set_collectives(self._h, self._cb_rs, self._cb_ag, None))

    def init_rccl(self, unique_id: bytes, rank, nranks):
        buf = C.create_string_buffer(unique_id, 128)
        self.check(lib().klr_ctx_init_rccl(self._h, buf, rank, nranks))

    def comm_info(self):
        r, n, c = C.c_int(0), C.c_int(0), C.c_int(0)
        self.check(lib().klr_ctx_comm_info(self._h, C.byref(r), C.byref(n), C.byref(c)))
        return dict(rank=r.value, nranks=n.value, rccl_ranks=c.value)

    def init_shm(self, region_addr, nbytes, rank, nranks):
        """Host-staged collectives through a shared region (klr_ctx_init_shm); region_addr = its address in this process."""
        self.check(lib().klr_ctx_init_shm(self._h, C.c_void_p(region_addr), int(nbytes), rank, nranks))

    def allreduce_host(self, arr):
        """Sums a float64 / int64 numpy array over the ranks in place (klr_allreduce_host)."""
        assert arr.dtype in (np.float64, np.int64) and arr.flags["C_CONTIGUOUS"]
        self.check(lib().klr_allreduce_host(self._h, _p(arr), arr.size, 1 if arr.dtype == np.int64 else 0))

    def comm_selfcheck(self):
        """Collective: runs every installed hook on small known buffers; raises when a result is wrong."""
        self.check(lib().klr_ctx_comm_selfcheck(self._h))

    def comm_stats(self):
        a, s_, f, c = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int(0)
        self.check(lib().klr_ctx_comm_stats(self._h, C.byref(a), C.byref(s_), C.byref(f), C.byref(c)))
        rg = C.c_int64(0)
        self.check(lib().klr_ctx_comm_ranged(self._h, C.byref(rg)))
        return dict(sweeps_allreduce=a.value, sweeps_scatter=s_.value, scatter_fallbacks=f.value, sweeps_ranged=rg.value, checked=bool(c.value))

    def comm_probe(self):
        n, rs, ag, ar, f, g = C.c_int64(0), C.c_double(0.0), C.c_double(0.0), C.c_double(0.0), C.c_int(0), C.c_int(0)
        self.check(lib().klr_ctx_comm_probe(self._h, C.byref(n), C.byref(rs), C.byref(ag), C.byref(ar), C.byref(f), C.byref(g)))
        return dict(elements=n.value, reduce_scatter_ms=rs.value, allgather_ms=ag.value, allreduce_ms=ar.value,
                    form={1: "reduce-scatter + all-gather", 0: "all-reduce", -1: "not measured"}[f.value], ranges=g.value)

    def set_option(self, name, value):
        """KLR_* switch of this context (value None unsets); contexts read the environment only when they are created."""
        self.check(lib().klr_ctx_set_option(self._h, name.encode(), None if value is None else str(value).encode()))

    def reload_env(self):
        self.check(lib().klr_ctx_reload_env(self._h))

    def allreduce_time(self, count, dtype=1, iters=20):
        ms = C.c_double(0.0)
        self.check(lib().klr_allreduce_time(self._h, int(count), int(dtype), int(iters), C.byref(ms)))
        return ms.value

    def prof_reserve(self, events):
        self.check(lib().klr_prof_reserve(self._h, int(events)))

    def prof_enable(self, on=True):
        self.check(lib().klr_prof_enable(self._h, int(on)))

    def prof_read(self, reset=True):
        n, ms = C.c_int64(0), C.c_double(0.0)
        self.check(lib().klr_prof_read(self._h, C.byref(n), C.byref(ms), int(reset)))
        return n.value, ms.value


    def prof_read_cd(self, reset=True):
        sw, nb, ms = C.c_int64(0), C.c_int64(0), C.c_double(0.0)
        self.check(lib().klr_prof_read_cd(self._h, C.byref(sw), C.byref(nb), C.byref(ms), int(reset)))
        return sw.value, nb.value, ms.value

    def evict(self, nbytes=1 << 30):
        self.check(lib().klr_prof_evict(self._h, int(nbytes)))


def version():
    return lib().klr_version().decode()


def kernel_hash():
    """Hash of the sweep / ingest sources the loaded library was built from (klr_version: "... kernels:<hash>")."""
    v = version()
    return v.split("kernels:")[1] if "kernels:" in v else ""


def rccl_unique_id():
    buf = C.create_string_buffer(128)
    rc = lib().klr_rccl_unique_id(buf)
    if rc != 0:
        raise KlrError("klr_rccl_unique_id failed: %s" % (lib().klr_last_error(None) or b"").decode())
    return buf.raw


ALPHABETS = {"nucleotide": 0, "gapped-nucleotide": 1, "iupac-nucleotide": 2}


class KmerClasses:
    """klr_kmer_classes: the k-mer equivalence classes of a counter (NewKmerCounter's arguments, kmerLr_learn.go:94)."""

    def __init__(self, M, N, revcomp=False, reverse=False, complement=False, alphabet="nucleotide", max_ambiguous=None):
        ma = None if max_ambiguous is None else (C.c_int * len(max_ambiguous))(*max_ambiguous)

        class Spec(C.Structure):
            _fields_ = [("M", C.c_int), ("N", C.c_int), ("complement", C.c_int), ("reverse", C.c_int), ("revcomp", C.c_int),
                        ("alphabet", C.c_int), ("n_max_ambiguous", C.c_int), ("max_ambiguous", C.c_void_p)]
        sp = Spec(M, N, int(complement), int(reverse), int(revcomp), ALPHABETS[alphabet], 0 if ma is None else len(max_ambiguous),
                  None if ma is None else C.cast(ma, C.c_void_p))
        self._h = C.c_void_p()
        err = C.create_string_buffer(256)
        if lib().klr_kmer_classes_create(C.byref(sp), C.byref(self._h), err, 256) != 0:
            raise KlrError("klr_kmer_classes_create: %s" % err.value.decode())

    def __len__(self):
        return lib().klr_kmer_classes_count(self._h)

    def name(self, cls):
        buf = C.create_string_buffer(128)
        if lib().klr_kmer_class_name(self._h, int(cls), buf, 128) != 0:
            raise KlrError("no class %d" % cls)
        return buf.value.decode()

    def find(self, kmer):
        return lib().klr_kmer_class_find(self._h, kmer.encode())

    def close(self):
        if self._h:
            lib().klr_kmer_classes_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Matrix:
    """klr_mat: rows [row0,row0+n) of the data set, resident in HBM as CSR + padded CSC."""

    def __init__(self, ctx, handle):
        self.ctx, self._h = ctx, handle
        n, dim, nnz, nt = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        ctx.check(lib().klr_mat_shape(self._h, C.byref(n), C.byref(dim), C.byref(nnz), C.byref(nt)))
        self.n, self.dim, self.nnz, self.n_total = n.value, dim.value, nnz.value, nt.value

    @staticmethod
    def from_csr(ctx, n, dim, rowptr, col, val, labels, n_total=0):
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        col = np.ascontiguousarray(col, dtype=np.int64)
        val = np.ascontiguousarray(val, dtype=np.float64)
        labels = np.ascontiguousarray(labels, dtype=np.uint8)
        h = C.c_void_p()
        ctx.check(lib().klr_mat_from_csr(ctx._h, n, dim, _p(rowptr), _p(col), _p(val), _p(labels), n_total, C.byref(h)))
        return Matrix(ctx, h)

    @staticmethod
    def from_sequences(ctx, seqs, labels, M=0, N=0, revcomp=False, binarize=False, n_total=0, classes=None):
        """seqs: list of str/bytes; k-mer counting and the whole matrix build run on the device.  classes: a KmerClasses
        (complement / reverse / gapped alphabet / max-ambiguous); otherwise plain or --revcomp classes of lengths M..N."""
        bs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
        off = np.zeros(len(bs) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(b) for b in bs])
        blob = b"".join(bs) or b"\0"
        labels = np.ascontiguousarray(labels, dtype=np.uint8)
        h = C.c_void_p()
        if classes is not None:
            ctx.check(lib().klr_mat_from_sequences_classes(ctx._h, C.c_char_p(blob), _p(off), len(bs), classes._h, int(binarize),
                                                           _p(labels), n_total, C.byref(h)))
            return Matrix(ctx, h)
        ctx.check(lib().klr_mat_from_sequences(ctx._h, C.c_char_p(blob), _p(off), len(bs), M, N, int(revcomp), int(binarize),
                                               _p(labels), n_total, C.byref(h)))
        return Matrix(ctx, h)

    def view(self, ctx):
        """klr_mat_view: this matrix for another context of the same device -- shared device storage, its own row mask.
        The matrix must outlive the view."""
        h = C.c_void_p()
        ctx.check(lib().klr_mat_view(ctx._h, self._h, C.byref(h)))
        v = Matrix(ctx, h)
        v._base = self
        return v

    def close(self):
        if self._h:
            lib().klr_mat_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def layout(self):
        t, g, vb = C.c_int(), C.c_int(), C.c_int()
        r, ns, pe = C.c_int64(), C.c_int64(), C.c_int64()
        self.ctx.check(lib().klr_mat_layout(self._h, C.byref(t), C.byref(r), C.byref(ns), C.byref(pe), C.byref(g),
                                            C.byref(vb)))
        fmt = C.c_int()
        sb, pb, tb = C.c_int64(), C.c_int64(), C.c_int64()
        self.ctx.check(lib().klr_mat_stream_info(self._h, C.byref(fmt), C.byref(sb), C.byref(pb), C.byref(tb)))
        names = {0: "u32", 1: "u24", 2: "r64", 3: "q64c12", 4: "q64c13"}
        kern = {0: "k_sweep_coo", 1: "k_sweep_coo", 2: "k_sweep_coo", 3: "k_sweep_q64", 4: "k_sweep_q64"}
        return dict(tiled=bool(t.value), tile_rows=r.value, nseg=ns.value, padded_entries=pe.value,
                    lanes_per_segment=g.value, value_bytes=vb.value, format=names.get(fmt.value, "csc"),
                    kernel=kern.get(fmt.value, "k_xt_s<%d>" % g.value), stream_bytes=sb.value, piece_bytes=pb.value,
                    table_bytes=tb.value)

    def to_csr(self):
        rowptr = np.zeros(self.n + 1, dtype=np.int64)
        col = np.zeros(max(self.nnz, 1), dtype=np.int64)
        val = np.zeros(max(self.nnz, 1), dtype=np.float64)
        self.ctx.check(lib().klr_mat_to_csr(self.ctx._h, self._h, _p(rowptr), _p(col), _p(val)))
        return rowptr, col[:self.nnz], val[:self.nnz]

    def set_active_rows(self, mask, n_active_total=0):
        """mask: uint8 per local row (None = all rows); n_active_total: active rows over all shards."""
        if mask is None:
            self.ctx.check(lib().klr_mat_set_active_rows(self.ctx._h, self._h, None, 0))
            self.n_total = self.n  # refreshed below
        else:
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
            assert mask.shape == (self.n,)
            self.ctx.check(lib().klr_mat_set_active_rows(self.ctx._h, self._h, _p(mask),
                                                         n_active_total or int(mask.sum())))
        n, dim, nnz, nt = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        self.ctx.check(lib().klr_mat_shape(self._h, C.byref(n), C.byref(dim), C.byref(nnz), C.byref(nt)))
        self.n_total = nt.value

    def class_weights(self):
        cw = np.zeros(2)
        self.ctx.check(lib().klr_class_weights(self.ctx._h, self._h, _p(cw)))
        return cw

    def select_columns(self, cols, scale=None, offset=None):
        cols = np.ascontiguousarray(cols, dtype=np.int64)
        scale, offset = _f64(scale), _f64(offset)
        h = C.c_void_p()
        self.ctx.check(lib().klr_mat_select_columns(self.ctx._h, self._h, _p(cols), len(cols), _p(scale), _p(offset),
                                                    C.byref(h)))
        return Matrix(self.ctx, h)


    def expand_cooccurrence(self):
        """Pair-feature matrix in CoeffIndex column order (klr_mat_expand_cooccurrence)."""
        h = C.c_void_p()
        self.ctx.check(lib().klr_mat_expand_cooccurrence(self.ctx._h, self._h, C.byref(h)))
        return Matrix(self.ctx, h)


def coeff_sub2ind(nb, i):
    a, b = C.c_int64(), C.c_int64()
    lib().klr_coeff_sub2ind(int(nb), int(i), C.byref(a), C.byref(b))
    return a.value, b.value


def gradient(ctx, mat, theta_idx, theta_val, cw=(1.0, 1.0), scale=None, offset=None):
    idx = np.ascontiguousarray(theta_idx, dtype=np.int64)
    tv = _f64(theta_val)
    cw, scale, offset = _f64(cw), _f64(scale), _f64(offset)
    g = np.zeros(mat.dim)
    ctx.check(lib().klr_gradient(ctx._h, mat._h, _p(idx), _p(tv), len(idx), _p(cw), _p(scale), _p(offset), _p(g)))
    return g


def gradient_dense(ctx, mat, theta, cw=(1.0, 1.0), scale=None, offset=None):
    theta = _f64(theta)
    idx = np.nonzero(theta)[0]
    return gradient(ctx, mat, idx, theta[idx], cw, scale, offset)


def loss(ctx, mat, theta, cw=(1.0, 1.0), lam=0.0):
    theta, cw = _f64(theta), _f64(cw)
    assert theta.shape == (mat.dim,)
    out = C.c_double(0.0)
    ctx.check(lib().klr_loss(ctx._h, mat._h, _p(theta), _p(cw), lam, C.byref(out)))
    return out.value


def log_pdf(ctx, mat, theta):
    theta = _f64(theta)
    out = np.zeros(mat.n)
    ctx.check(lib().klr_log_pdf(ctx._h, mat._h, _p(theta), _p(out)))
    return out


TRANSFORMS = {"standardizer": 1, "variance-scaler": 2, "max-abs-scaler": 3, "mean-scaler": 4}


def transform_fit(ctx, mat, kind):
    """TransformFull.Fit on the device: returns (scale, offset or None)."""
    scale, offset = np.zeros(mat.dim), np.zeros(mat.dim)
    ho = C.c_int(0)
    ctx.check(lib().klr_transform_fit(ctx._h, mat._h, TRANSFORMS[kind], _p(scale), _p(offset), C.byref(ho)))
    return scale, (offset if ho.value else None)


def select(ctx, g, b, c_prev, N, lambda_prev=0.0, eps_lambda=0.0):
    g = _f64(g)
    b = np.array(b, dtype=np.uint8, copy=True)
    lam, c, ok = C.c_double(0.0), C.c_int64(0), C.c_int(0)
    ctx.check(lib().klr_select(ctx._h, _p(g), len(g), _p(b), c_prev, N, lambda_prev, eps_lambda,
                               C.byref(lam), C.byref(c), C.byref(ok)))
    return bool(ok.value), lam.value, c.value, b


def select_fixed(ctx, g_full, b, c_prev, lam, max_features=0):
    g_full = _f64(g_full)
    b = np.array(b, dtype=np.uint8, copy=True)
    c, ok = C.c_int64(0), C.c_int(0)
    ctx.check(lib().klr_select_fixed(ctx._h, _p(g_full), len(g_full) - 1, _p(b), c_prev, lam, max_features,
                                     C.byref(c), C.byref(ok)))
    return bool(ok.value), c.value, b


def sweep_select(ctx, full, reduced, red_cols, theta_red, cw, N, lambda_prev=0.0, eps_lambda=0.0, want_g=False,
                 scale=None, offset=None):
    """Returns dict(ok, lam, c, b, loss, g)."""
    theta_red, cw, scale, offset = _f64(theta_red), _f64(cw), _f64(scale), _f64(offset)
    red_cols = None if red_cols is None else np.ascontiguousarray(red_cols, dtype=np.int64)
    b = np.zeros(full.dim, dtype=np.uint8)
    g = np.zeros(full.dim) if want_g else None
    lam, c, ok, ls = C.c_double(0.0), C.c_int64(0), C.c_int(0), C.c_double(0.0)
    ctx.check(lib().klr_sweep_select(ctx._h, full._h, None if reduced is None else reduced._h, _p(red_cols),
                                     _p(theta_red), len(theta_red), _p(cw), N, lambda_prev, eps_lambda,
                                     _p(scale), _p(offset), _p(b),
                                     C.byref(lam), C.byref(c), C.byref(ok), C.byref(ls), _p(g)))
    return dict(ok=bool(ok.value), lam=lam.value, c=c.value, b=b, loss=ls.value, g=g)


class SweepStep:
    """klr_sweep_select with every argument marshalled once: the per-call Python overhead of the generic
    wrapper (array conversions, a fresh dim-sized mask) is of the order of the GPU work at small shapes."""

    def __init__(self, ctx, full, reduced, red_cols, theta_red, cw, N, lambda_prev=0.0, eps_lambda=0.0):
        self.ctx, self.full, self.reduced = ctx, full, reduced
        self.theta = np.ascontiguousarray(theta_red, dtype=np.float64)
        self.cols = None if red_cols is None else np.ascontiguousarray(red_cols, dtype=np.int64)
        self.cw = np.ascontiguousarray(cw, dtype=np.float64)
        self.b = np.zeros(full.dim, dtype=np.uint8)
        self.lam, self.c, self.ok, self.loss = C.c_double(0.0), C.c_int64(0), C.c_int(0), C.c_double(0.0)
        self._fn = lib().klr_sweep_select
        self._args = (ctx._h, full._h, None if reduced is None else reduced._h, _p(self.cols), _p(self.theta),
                      C.c_int64(len(self.theta)), _p(self.cw), C.c_int64(N), C.c_double(lambda_prev), C.c_double(eps_lambda),
                      None, None, _p(self.b), C.byref(self.lam), C.byref(self.c), C.byref(self.ok), C.byref(self.loss), None)

    def run(self):
        rc = self._fn(*self._args)
        if rc != 0:
            self.ctx.check(rc)
        return self.lam.value

    def result(self):
        return dict(ok=bool(self.ok.value), lam=self.lam.value, c=self.c.value, b=self.b, loss=self.loss.value, g=None)


def cd_solve(ctx, red, theta, cw=(1.0, 1.0), l1reg=0.0, l2reg=0.0, eps=0.0, eps_loss=1e-8, eval_loss=True,
             max_iter=0, recs_cap=100000):
    theta = np.array(theta, dtype=np.float64, copy=True)
    cw = _f64(cw)
    assert theta.shape == (red.dim,)
    recs = (IterRec * recs_cap)()
    n_recs = C.c_int64(0)
    ctx.check(lib().klr_cd_solve(ctx._h, red._h, _p(theta), _p(cw), l1reg, l2reg, eps, eps_loss, int(eval_loss),
                                 max_iter, C.cast(recs, C.c_void_p), recs_cap, C.byref(n_recs)))
    out = [dict(iteration=r.iteration, nonzero=r.nonzero, change=r.change, lambda_=r.lambda_, loss=r.loss,
                seconds=r.seconds) for r in recs[:min(n_recs.value, recs_cap)]]
    return theta, out


def cd_sweep_rate(ctx, red, cw, l1reg, sweeps=64):
    """Coordinate-descent sweeps per second from the device timestamps of the sweep kernels themselves: `sweeps`
    soft-threshold sweeps inside ONE re-weighting (eps = eps_loss = 0: nothing stops them), timed by the HIP events
    klr_cd_solve puts around every batch while profiling is on.  Best of 3."""
    best = None
    for _ in range(3):
        ctx.prof_enable(True)
        ctx.prof_read_cd(reset=True)
        cd_solve(ctx, red, np.zeros(red.dim), cw=cw, l1reg=l1reg, eps=0.0, eps_loss=0.0, eval_loss=True, max_iter=sweeps)
        sw, nb, ms = ctx.prof_read_cd(reset=True)
        ctx.prof_read(reset=True)
        ctx.prof_enable(False)
        if sw > 0 and ms > 0 and (best is None or ms / sw < best[2] / best[0]):
            best = (sw, nb, ms)
    if best is None:
        return dict(sweeps_per_s=0.0, sweeps=0, batches=0, device_ms=0.0, coordinates=int(red.dim))
    return dict(sweeps_per_s=round(best[0] / (best[2] * 1e-3), 1), sweeps=best[0], batches=best[1], device_ms=round(best[2], 4),
                us_per_sweep=round(best[2] * 1e3 / best[0], 3), coordinates=int(red.dim),
                how="HIP events around the CD sweep kernels of each batch (klr_prof_read_cd)")


def synth_csr(seed, row0, n_local, n_total, m, draws, S, threads=0):
    """Host-side synthetic generator of the product library (no GPU needed)."""
    rowptr = np.zeros(n_local + 1, dtype=np.int64)
    col = np.zeros(max(n_local * (draws + 1), 1), dtype=np.int64)
    val = np.zeros(max(n_local * (draws + 1), 1), dtype=np.float64)
    labels = np.zeros(max(n_local, 1), dtype=np.uint8)
    rc = lib().klr_synth_csr(C.c_uint64(seed), row0, n_local, n_total, m, draws, S, _p(rowptr), _p(col), _p(val),
                             _p(labels), threads)
    if rc != 0:
        raise KlrError("klr_synth_csr failed: %s" % (lib().klr_last_error(None) or b"").decode())
    nnz = int(rowptr[-1])
    return rowptr, col[:nnz], val[:nnz], labels[:n_local]
