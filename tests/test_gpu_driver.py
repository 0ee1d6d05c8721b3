"""GPU: the host driver over the HIP backend must follow the same leapfrog path as over the oracle
backend; the all-reduce hook plumbing (f64 and int64) is exercised with an emulated second rank."""
import ctypes as C
import os

import numpy as np
import pytest

import kmerlr_amd as K
from oracle import oracle as O
from tests import oracle_driver as OD

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = K.Context(0)
    yield c
    c.close()


def run_both(ctx, x, labels, lambda_auto, scale=None, **cfg):
    m = K.Matrix.from_csr(ctx, x.n, x.dim, x.rowptr, x.col, x.val, labels)
    eg = K.hip_estimator(ctx, **cfg)
    cg = eg.estimate(m._h, lambda_auto, scale=scale)
    h = OD.matrix(x, labels)
    eo = OD.estimator(**cfg)
    co = eo.estimate(h, lambda_auto, scale=scale)
    OD.free_matrix(h)
    return eg, cg, eo, co


def assert_same_path(eg, cg, eo, co):
    lg, ng = eg.epochs(); lo, no = eo.epochs()
    assert list(ng) == list(no)                                     # identical path sequence
    np.testing.assert_allclose(lg, lo, rtol=1e-6)
    assert len(cg) == len(co)
    for a, b in zip(cg, co):
        np.testing.assert_array_equal(a["cols"], b["cols"])         # identical selected feature indices
        np.testing.assert_allclose(a["theta"], b["theta"], rtol=1e-6, atol=1e-10)


def test_readme_scores_on_gpu(ctx, scores_data):
    x, labels, _ = scores_data
    eg, cg, eo, co = run_both(ctx, x, labels, [2, 5], save_trace=1, save_path=1)
    assert_same_path(eg, cg, eo, co)
    assert "%e" % eg.epochs()[0][0] == "2.496875e+00"
    assert list(cg[0]["cols"]) == [0, 2, 7]
    tg, to = eg.trace(), eo.trace()
    assert [r["iteration"] for r in tg] == [r["iteration"] for r in to]
    assert [r["nonzero"] for r in tg] == [r["nonzero"] for r in to]


def test_synthetic_leapfrog_path_on_gpu(ctx):
    x, labels = O.synth_csr(20261004, 0, 5000, 5000, 4096, 300, 40)      # tiled count-data layout
    eg, cg, eo, co = run_both(ctx, x, labels, [5, 20], balance=1)
    assert_same_path(eg, cg, eo, co)
    assert len(cg[1]["cols"]) == 21


def test_leapfrog_path_large_selections_on_gpu(ctx):
    """The same path check where the reduced problems leave the LDS-resident solver: 151 coordinates (Gram rows from
    L2, two steps ahead) and 601 (blocked sweep over 16 wavefronts), the sub-select binned over up to 1024 columns."""
    x, labels = O.synth_csr(17, 0, 3000, 3000, 8192, 400, 1200)
    # integer column sums tie by the hundred at 3000 rows, and the midpoint rule for lambda is discontinuous at a tie
    # ("Selection at ties"): a generic per-column scale makes every |g_j| distinct
    scale = np.r_[1.0, 1.0 + 0.5 * np.random.default_rng(5).random(x.dim - 1)]
    eg, cg, eo, co = run_both(ctx, x, labels, [150, 600], scale=scale, max_iterations=4000)
    assert_same_path(eg, cg, eo, co)
    assert len(cg[0]["cols"]) >= 151 and len(cg[1]["cols"]) >= 601


def test_fixed_lambda_and_scale_transform_on_gpu(ctx):
    x, labels = O.synth_csr(3, 0, 1500, 1500, 600, 100, 10)
    g = O.gradient(x, labels, np.zeros(x.dim))
    lam = float(np.sort(np.abs(g[1:]))[-8])
    eg, cg, eo, co = run_both(ctx, x, labels, [0], lambda_=lam, max_features=6)
    assert_same_path(eg, cg, eo, co)
    d = x.to_dense()
    sd = d[:, 1:].std(axis=0, ddof=1); sd[sd == 0] = 1.0
    scale = np.r_[1.0, 1.0 / sd]
    eg, cg, eo, co = run_both(ctx, x, labels, [6], scale=scale)
    assert_same_path(eg, cg, eo, co)


def test_allreduce_hook_f64_and_int64(ctx):
    """Emulated second rank: the hook doubles the buffer (as if an identical shard had been added) while
    n_total = 2n; the results must equal the single-rank ones.  Checks that the hook sees every reduction
    of gradient, loss, Gram and class counts, in both element types."""
    hip = C.CDLL("libamdhip64.so")
    seen = []

    def hook(buf, count, dtype, stream):
        hip.hipStreamSynchronize(C.c_void_p(stream))
        host = np.empty(count, dtype=np.int64 if dtype == 1 else np.float64)
        assert hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), C.c_void_p(buf), C.c_size_t(count * 8), 2) == 0
        host *= 2
        assert hip.hipMemcpy(C.c_void_p(buf), host.ctypes.data_as(C.c_void_p), C.c_size_t(count * 8), 1) == 0
        seen.append(dtype)

    for shape in ((5000, 5000, 4096, 300, 24), (700, 700, 300, 60, 8)):           # tiled (int64) and plain (f64) layouts
        x, labels = O.synth_csr(11, 0, *shape)
        cw = O.class_weights(labels)
        base = K.Context(0)
        m1 = K.Matrix.from_csr(base, x.n, x.dim, x.rowptr, x.col, x.val, labels)
        r1 = K.sweep_select(base, m1, None, None, np.zeros(1), cw, 10, want_g=True)
        c2 = K.Context(0)
        c2.set_allreduce(hook, 0, 2)
        m2 = K.Matrix.from_csr(c2, x.n, x.dim, x.rowptr, x.col, x.val, labels, n_total=2 * x.n)
        np.testing.assert_allclose(m2.class_weights(), cw)
        r2 = K.sweep_select(c2, m2, None, None, np.zeros(1), cw, 10, want_g=True)
        np.testing.assert_allclose(r2["g"], r1["g"], rtol=1e-12, atol=1e-18)
        np.testing.assert_array_equal(r2["b"], r1["b"])
        assert abs(r2["loss"] - r1["loss"]) < 1e-12
        cols = np.nonzero(r1["b"])[0]
        t1, _ = K.cd_solve(base, m1.select_columns(cols), np.zeros(len(cols)), cw=cw, l1reg=r1["lam"] * x.n)
        t2, _ = K.cd_solve(c2, m2.select_columns(cols), np.zeros(len(cols)), cw=cw, l1reg=r1["lam"] * 2 * x.n)
        np.testing.assert_allclose(t2, t1, rtol=1e-6, atol=1e-12)
        m1.close(); m2.close(); c2.close(); base.close()
    assert 0 in seen and 1 in seen


@pytest.mark.parametrize("N", [150, 600])
def test_allreduce_hook_large_selection(ctx, N, setopt):
    """The same emulated second rank for the solver paths of larger selections: tiled Gram partials reduced to the
    integer Gram, all-reduced and assembled around the hook (128 < d <= 1024), q = G theta carried between batches,
    the wave (d <= 512) and blocked (d > 512) sweeps."""
    hip = C.CDLL("libamdhip64.so")

    def hook(buf, count, dtype, stream):
        hip.hipStreamSynchronize(C.c_void_p(stream))
        host = np.empty(count, dtype=np.int64 if dtype == 1 else np.float64)
        assert hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), C.c_void_p(buf), C.c_size_t(count * 8), 2) == 0
        host *= 2
        assert hip.hipMemcpy(C.c_void_p(buf), host.ctypes.data_as(C.c_void_p), C.c_size_t(count * 8), 1) == 0

    setopt("KLR_GRAM_PATH", "lds")
    x, labels = O.synth_csr(13, 0, 3000, 3000, 8192, 400, 2 * N)
    cw = O.class_weights(labels)
    base = K.Context(0)
    m1 = K.Matrix.from_csr(base, x.n, x.dim, x.rowptr, x.col, x.val, labels)
    r1 = K.sweep_select(base, m1, None, None, np.zeros(1), cw, N)
    cols = np.nonzero(r1["b"])[0]
    c2 = K.Context(0)
    c2.set_allreduce(hook, 0, 2)
    m2 = K.Matrix.from_csr(c2, x.n, x.dim, x.rowptr, x.col, x.val, labels, n_total=2 * x.n)
    t1, rec1 = K.cd_solve(base, m1.select_columns(cols), np.zeros(len(cols)), cw=cw, l1reg=r1["lam"] * x.n, max_iter=40)
    t2, rec2 = K.cd_solve(c2, m2.select_columns(cols), np.zeros(len(cols)), cw=cw, l1reg=r1["lam"] * 2 * x.n, max_iter=40)
    assert len(rec1) == len(rec2)
    np.testing.assert_allclose(t2, t1, rtol=1e-6, atol=1e-10)
    m1.close(); m2.close(); c2.close(); base.close()


def test_active_row_masks_on_gpu(ctx):
    """klr_mat_set_active_rows: masked sums equal the oracle on the filtered rows (tiled and plain layouts)."""
    for shape in ((6000, 6000, 4096, 250, 16), (500, 500, 300, 60, 8)):
        x, labels = O.synth_csr(31, 0, *shape)
        rng = np.random.default_rng(4)
        mask = (rng.uniform(size=x.n) < 0.7).astype(np.uint8)
        tr = np.nonzero(mask)[0]
        rp = np.concatenate([[0], np.cumsum((x.rowptr[1:] - x.rowptr[:-1])[tr])])
        idx = np.concatenate([np.arange(x.rowptr[i], x.rowptr[i + 1]) for i in tr])
        xs = O.Csr(len(tr), x.dim, rp, x.col[idx], x.val[idx])
        m = K.Matrix.from_csr(ctx, x.n, x.dim, x.rowptr, x.col, x.val, labels)
        m.set_active_rows(mask)
        assert m.n_total == len(tr)
        cw = O.class_weights(labels[tr])
        np.testing.assert_array_equal(m.class_weights(), cw)
        theta = np.zeros(x.dim); theta[0] = 0.2; theta[[3, 40, 200]] = [0.1, -0.2, 0.05]
        g = K.gradient_dense(ctx, m, theta, cw=cw); go = O.gradient(xs, labels[tr], theta, cw=cw)
        assert np.max(np.abs(g - go)) <= 1e-11 * np.max(np.abs(go))
        assert abs(K.loss(ctx, m, theta, cw=cw) - O.loss(xs, labels[tr], theta, cw=cw)) < 1e-12
        r = K.sweep_select(ctx, m, None, None, np.zeros(1), cw, 8)
        b = np.zeros(x.dim, dtype=np.uint8); b[0] = 1
        ok, lam, c, b2 = O.select(O.gradient(xs, labels[tr], np.zeros(x.dim), cw=cw)[1:], b, 0, 8)
        np.testing.assert_array_equal(r["b"], b2)
        cols = np.nonzero(b2)[0]
        red = m.select_columns(cols)                                   # inherits the mask
        th, _ = K.cd_solve(ctx, red, np.zeros(len(cols)), cw=cw, l1reg=r["lam"] * len(tr))
        tho, _ = O.cd_solve(O.select_columns(xs, cols), labels[tr], np.zeros(len(cols)), cw=cw, l1reg=lam * len(tr))
        np.testing.assert_allclose(th, tho, rtol=1e-6, atol=1e-12)
        m.set_active_rows(None)
        assert m.n_total == x.n
        assert np.max(np.abs(K.gradient_dense(ctx, m, theta) - O.gradient(x, labels, theta))) <= 1e-11
        m.close()


def test_crossvalidation_on_gpu_matches_oracle_driver(ctx):
    x, labels = O.synth_csr(41, 0, 3000, 3000, 2048, 200, 20)
    m = K.Matrix.from_csr(ctx, x.n, x.dim, x.rowptr, x.col, x.val, labels)
    rg = K.hip_estimator(ctx, balance=1).crossvalidate(m._h, x.n, 5, [4, 10], seed=1)
    h = OD.matrix(x, labels)
    ro = OD.estimator(balance=1).crossvalidate(h, x.n, 5, [4, 10], seed=1)
    OD.free_matrix(h)
    np.testing.assert_array_equal(rg["groups"], ro["groups"])
    for a, b in zip(rg["classifiers"], ro["classifiers"]):
        np.testing.assert_array_equal(a["cols"], b["cols"])
        np.testing.assert_allclose(a["theta"], b["theta"], rtol=1e-6, atol=1e-10)
    np.testing.assert_allclose(rg["predictions"], ro["predictions"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(rg["loss_train"], ro["loss_train"], rtol=1e-8)
    np.testing.assert_allclose(rg["loss_test"], ro["loss_test"], rtol=1e-8)


@pytest.mark.parametrize("kw", [dict(), dict(validation_size=0.2, data_transform=1), dict(ensemble_size=2)])
def test_crossvalidation_folds_in_threads_on_gpu(ctx, kw):
    """klr_config.threads_cv: three folds at once, each on a forked context and a view of the data (shared device storage,
    its own row mask); every number equals the run with the folds in turn."""
    x, labels = O.synth_csr(43, 0, 3000, 3000, 2048, 200, 20)
    m = K.Matrix.from_csr(ctx, x.n, x.dim, x.rowptr, x.col, x.val, labels)
    res = []
    for threads in (1, 3):
        est = K.hip_estimator(ctx, balance=1, threads_cv=threads, save_trace=1, **kw)
        r = est.crossvalidate(m._h, x.n, 5, [4, 10], seed=2)
        r["trace"] = [(t["iteration"], t["nonzero"], t["loss"]) for t in est.trace()]
        r["folds"] = [[c["theta"] for c in est.fold(f).classifiers()] for f in range(5)]
        res.append(r)
        est.close()
    a, b = res
    np.testing.assert_array_equal(a["predictions"], b["predictions"])
    np.testing.assert_array_equal(a["loss_train"], b["loss_train"])
    np.testing.assert_array_equal(a["loss_test"], b["loss_test"])
    assert len(a["classifiers"]) == len(b["classifiers"]) > 0
    for ca, cb in zip(a["classifiers"], b["classifiers"]):
        np.testing.assert_array_equal(ca["cols"], cb["cols"])
        np.testing.assert_array_equal(ca["theta"], cb["theta"])
    assert a["trace"] == b["trace"]
    for fa, fb in zip(a["folds"], b["folds"]):
        for u, v in zip(fa, fb):
            np.testing.assert_array_equal(u, v)
    # the matrix is untouched by the views: a plain estimate afterwards sees all rows
    assert m.n_total == x.n
    m.close()


def test_ensemble_members_in_threads_on_gpu(ctx):
    """klr_config.threads: the members of an ensemble at once on forked contexts; the same classifiers as in turn."""
    x, labels = O.synth_csr(47, 0, 3000, 3000, 2048, 200, 20)
    m = K.Matrix.from_csr(ctx, x.n, x.dim, x.rowptr, x.col, x.val, labels)
    res = []
    for threads in (1, 3):
        est = K.hip_estimator(ctx, threads=threads)
        res.append(est.estimate_ensemble(m._h, x.n, 4, [4, 10], seed=2))
        est.close()
    assert len(res[0]) == len(res[1]) == 8
    for a, b in zip(*res):
        np.testing.assert_array_equal(a["cols"], b["cols"])
        np.testing.assert_array_equal(a["theta"], b["theta"])
    assert m.n_total == x.n
    m.close()


def test_full_size_estimate_kkt_and_reproducibility(ctx, setopt):
    """The whole leapfrog run at BASELINE configs[2] size (100k x 65,536, --lambda-auto=100), without the oracle (it would
    take minutes): the classifier has exactly N nonzero coefficients, satisfies the KKT conditions of the L1 problem at
    its lambda (|g_j| = lambda with the opposite sign on the support, bias gradient 0) to the accuracy the loss rule stops
    at, is bit-identical from run to run, and does not depend on how the sweeps are batched."""
    from kmerlr_amd import _driver
    n, m, N = 100000, 65536, 100
    rowptr, col, val, labels = K.synth_csr(20261004, 0, n, n, m, 993, 2 * N)
    full = K.Matrix.from_csr(ctx, n, m + 1, rowptr, col, val, labels, n)

    def run():
        est = _driver.hip_estimator(ctx)
        est.estimate(full._h, [N])
        cl = est.classifiers()[0]
        ep = est.epochs()
        est.close()
        return cl, ep

    cl, ep = run()
    cols, theta, lam = cl["cols"], cl["theta"], cl["lam"]
    assert int(np.count_nonzero(theta[1:])) == N and len(cols) >= N + 1
    red = full.select_columns(cols)
    g = K.gradient_dense(ctx, red, theta, full.class_weights())
    nz = theta[1:] != 0
    assert abs(g[0]) < 1e-5
    assert np.max(np.abs(np.abs(g[1:][nz]) - lam)) < 1e-3 * lam
    assert np.all(np.sign(g[1:][nz]) == -np.sign(theta[1:][nz]))
    if (~nz).any():
        assert np.max(np.abs(g[1:][~nz])) <= lam * (1 + 1e-3)
    cl2, ep2 = run()
    np.testing.assert_array_equal(cl2["cols"], cols)
    np.testing.assert_array_equal(cl2["theta"], theta)
    assert ep2 == ep
    setopt("KLR_CD_BATCH", "3")
    setopt("KLR_CD_FIRST_BATCH", "8")
    cl3, ep3 = run()
    np.testing.assert_array_equal(cl3["cols"], cols)
    np.testing.assert_array_equal(cl3["theta"], theta)
    red.close(); full.close()


def test_rccl_collectives_single_rank_plumbing(ctx):
    """The library's own RCCL communicator with ONE rank (two ranks on one card are refused by RCCL; more cards are not on
    the test box): all-reduce, and -- forced -- the reduce-scatter + all-gather form of the sweep's collective, run through
    ncclAllReduce / ncclReduceScatter / ncclAllGather and must reproduce the hook-free results bit for bit."""
    import subprocess
    import sys
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
import kmerlr_amd as K
n, m, N = 6000, 30000, 20
rowptr, col, val, labels = K.synth_csr(20261004, 0, n, n, m, 300, 2 * N)
base = K.Context(0)
mb = K.Matrix.from_csr(base, n, m + 1, rowptr, col, val, labels)
cw = mb.class_weights()
r0 = K.sweep_select(base, mb, None, None, np.zeros(1), cw, N)
ctx = K.Context(0)
ctx.init_rccl(K.rccl_unique_id(), 0, 1)
info = ctx.comm_info()
assert info["rccl_ranks"] == 1 and info["nranks"] == 1, info
mc = K.Matrix.from_csr(ctx, n, m + 1, rowptr, col, val, labels)
r1 = K.sweep_select(ctx, mc, None, None, np.zeros(1), cw, N)           # KLR_RS_FORCE: reduce-scatter + all-gather
assert r1["lam"] == r0["lam"] and r1["loss"] == r0["loss"] and r1["c"] == r0["c"] and np.array_equal(r1["b"], r0["b"])
for G in (1, 2, 3, 4):                                                  # strip ranges of the overlapped form, RCCL on a second stream
    ctx.set_option("KLR_SWEEP_RANGES", G)
    rg = K.sweep_select(ctx, mc, None, None, np.zeros(1), cw, N)
    assert rg["lam"] == r0["lam"] and rg["loss"] == r0["loss"] and rg["c"] == r0["c"] and np.array_equal(rg["b"], r0["b"]), G
ctx.set_option("KLR_SWEEP_RANGES", None)
assert ctx.comm_stats()["sweeps_scatter"] >= 5
r2 = K.sweep_select(ctx, mc, None, None, np.zeros(1), cw, N, want_g=True)   # g_out: the all-reduce form
assert r2["lam"] == r0["lam"] and np.array_equal(r2["b"], r0["b"])
assert ctx.allreduce_time(m + 6, 1, 3) >= 0.0
print("rccl single rank ok")
''' % ROOT
    env = dict(os.environ, KLR_RS_FORCE="1", KLR_RS_MIN_DIM="1")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "rccl single rank ok" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]
