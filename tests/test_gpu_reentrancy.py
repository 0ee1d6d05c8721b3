"""GPU: one klr_ctx per concurrently running estimator (include/kmerlr.h; the reference runs one estimator per CV fold
/ ensemble member in parallel: kmerLr_crossvalidation.go:176, kmerLr_estimator_ensemble.go:81).  Two host threads, a
context each, run whole leapfrog estimates at the same time on different matrices: the results must be bitwise those
of the same runs one after the other."""
import threading

import numpy as np
import pytest

import kmerlr_amd as K
from kmerlr_amd import _driver

pytestmark = pytest.mark.gpu

SHAPES = ((20261004, 6000, 4096, 300, 16, [6, 16]),       # packed integer stream (n >= 2048, nnz >= 2^20)
          (7, 1500, 700, 90, 10, [4, 10]))                # plain CSC, f64 protocol


def job(shape, out, key):
    seed, n, m, draws, N, la = shape
    ctx = K.Context(0)
    rowptr, col, val, labels = K.synth_csr(seed, 0, n, n, m, draws, 2 * N)
    mat = K.Matrix.from_csr(ctx, n, m + 1, rowptr, col, val, labels)
    res = []
    for _ in range(2):
        est = _driver.hip_estimator(ctx, save_trace=1)
        cls = est.estimate(mat._h, la)
        lam, nz = est.epochs()
        res.append((cls, lam, nz, len(est.trace())))
        est.close()
    mat.close(); ctx.close()
    out[key] = res


def same(a, b):
    (ca, la, na, ta), (cb, lb, nb, tb) = a, b
    assert la == lb and na == nb and ta == tb and len(ca) == len(cb)
    for x, y in zip(ca, cb):
        np.testing.assert_array_equal(x["cols"], y["cols"])
        np.testing.assert_array_equal(x["theta"], y["theta"])
        assert x["lam"] == y["lam"]


def test_two_contexts_two_threads():
    seq, par = {}, {}
    for i, sh in enumerate(SHAPES):
        job(sh, seq, i)
    for rep in range(2):
        ths = [threading.Thread(target=job, args=(sh, par, i)) for i, sh in enumerate(SHAPES)]
        for t in ths:
            t.start()
        for t in ths:
            t.join(timeout=600)
        assert len(par) == len(SHAPES), "a thread did not finish (or raised)"
        for i in range(len(SHAPES)):
            same(seq[i][0], seq[i][1])
            same(seq[i][0], par[i][0]); same(seq[i][0], par[i][1])
        par.clear()


def test_options_belong_to_the_context(monkeypatch):
    """A context reads the KLR_* switches once, at creation; klr_ctx_set_option changes one context only."""
    monkeypatch.setenv("KLR_ENTRY_FORMAT", "u32")
    a = K.Context(0)
    monkeypatch.delenv("KLR_ENTRY_FORMAT")
    b = K.Context(0)
    rowptr, col, val, labels = K.synth_csr(3, 0, 4096, 4096, 4096, 300, 12)
    ma = K.Matrix.from_csr(a, 4096, 4097, rowptr, col, val, labels)
    mb = K.Matrix.from_csr(b, 4096, 4097, rowptr, col, val, labels)
    assert ma.layout()["format"] == "u32" and mb.layout()["format"].startswith("q64")
    b.set_option("KLR_ENTRY_FORMAT", "r64")
    mc = K.Matrix.from_csr(b, 4096, 4097, rowptr, col, val, labels)
    assert mc.layout()["format"] == "r64"
    b.set_option("KLR_ENTRY_FORMAT", None)
    md = K.Matrix.from_csr(b, 4096, 4097, rowptr, col, val, labels)
    assert md.layout()["format"].startswith("q64")
    cw = (1.0, 1.0)
    g = [K.sweep_select(c, m_, None, None, np.zeros(1), cw, 8, want_g=True)["g"] for c, m_ in ((a, ma), (b, mb), (b, mc), (b, md))]
    for x in g[1:]:
        np.testing.assert_array_equal(g[0], x)                       # exact integer sums whatever the layout
    for m_ in (ma, mb, mc, md):
        m_.close()
    a.close(); b.close()


def test_forked_context_and_matrix_view():
    """klr_ctx_fork / klr_mat_view: a view shares the device storage of the matrix and has its own row mask -- the base
    matrix keeps its rows, two views with different masks work side by side from two threads, and a released context
    comes back warm on the next fork."""
    ctx = K.Context(0)
    n, mm = 6000, 4096
    rowptr, col, val, labels = K.synth_csr(61, 0, n, n, mm, 300, 10)
    m = K.Matrix.from_csr(ctx, n, mm + 1, rowptr, col, val, labels)
    theta = np.zeros(mm + 1)
    g_all = K.gradient_dense(ctx, m, theta)
    rng = np.random.default_rng(3)
    masks = [(rng.random(n) < 0.6).astype(np.uint8) for _ in range(2)]
    want = []
    for mk in masks:
        m.set_active_rows(mk, int(mk.sum()))
        want.append(K.gradient_dense(ctx, m, theta))
    m.set_active_rows(None)
    got = [None, None]
    forks = [ctx.fork(), ctx.fork()]

    def work(t):
        v = m.view(forks[t])
        v.set_active_rows(masks[t], int(masks[t].sum()))
        for _ in range(5):
            got[t] = K.gradient_dense(forks[t], v, theta)
        v.close()

    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    for t in range(2):
        np.testing.assert_array_equal(got[t], want[t])
    np.testing.assert_array_equal(K.gradient_dense(ctx, m, theta), g_all)        # the base never saw a mask
    handles = {forks[0]._h.value, forks[1]._h.value}
    forks[0].release(); forks[1].release()
    again = ctx.fork()
    assert again._h.value in handles                                             # a pooled context comes back
    again.release()
    m.close(); ctx.close()
