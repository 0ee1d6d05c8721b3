"""Parity of the HIP path (through the C ABI) with the CPU oracle on the same seeded inputs.
Integer / index results must be bit-exact; floating point within the tolerances written here
(north_star: coefficients within 1e-6 relative; gradient / loss far tighter)."""
import numpy as np
import pytest

import kmerlr_amd as K
from oracle import oracle as O

pytestmark = pytest.mark.gpu

G_RTOL = 1e-11      # gradient entries, relative to max|g|
LOSS_ATOL = 1e-12
THETA_RTOL = 1e-6   # north_star tolerance for coefficients


@pytest.fixture(scope="module")
def ctx():
    c = K.Context(0)
    yield c
    c.close()


def upload(ctx, x, labels, n_total=0):
    return K.Matrix.from_csr(ctx, x.n, x.dim, x.rowptr, x.col, x.val, labels, n_total)


def assert_g_close(g, g_ref):
    scale = max(np.max(np.abs(g_ref)), 1e-300)
    assert np.max(np.abs(g - g_ref)) <= G_RTOL * scale


@pytest.fixture(scope="module")
def synth_small():
    x, labels = O.synth_csr(20261004, 0, 600, 600, 2048, 300, 16)
    return x, labels


@pytest.fixture(scope="module")
def scores(scores_data):
    return scores_data


def test_golden_g1_g2_g3_on_gpu(ctx, scores):
    x, labels, _ = scores
    m = upload(ctx, x, labels)
    g = K.gradient(ctx, m, [], [], cw=(1.0, 1.0))
    np.testing.assert_allclose(g[1:], [0.06875, 3.984375, -1.009375, 0.0228125, 0.96875, 0.045625, 4.03125],
                               rtol=0, atol=1e-13)
    b = np.zeros(x.dim, dtype=np.uint8); b[0] = 1
    ok, lam, c, b2 = K.select(ctx, g[1:], b, 0, 2)
    assert ok and c == 2 and list(np.nonzero(b2)[0]) == [0, 2, 7]
    assert "%e" % lam == "2.496875e+00"
    red = m.select_columns([0, 2, 7])
    theta = np.array([0.842178566751775, -0.05466291047449, -0.03026279836545])
    assert abs(K.loss(ctx, red, theta, lam=4.647556) - 0.813659729805629) < 1e-12


def test_roundtrip_and_select_columns(ctx, synth_small):
    x, labels = synth_small
    m = upload(ctx, x, labels)
    rp, col, val = m.to_csr()
    np.testing.assert_array_equal(rp, x.rowptr)
    np.testing.assert_array_equal(col, x.col)
    np.testing.assert_array_equal(val, x.val)
    rng = np.random.default_rng(3)
    cols = np.concatenate([[0], 1 + np.sort(rng.choice(x.dim - 1, size=40, replace=False))])
    scale = rng.uniform(0.5, 2.0, size=len(cols)); scale[0] = 1.0
    off = rng.uniform(-0.3, 0.3, size=len(cols)); off[0] = 0.0
    for sc, of in ((None, None), (scale, None), (scale, off), (None, off)):
        red = m.select_columns(cols, scale=sc, offset=of)
        ro = O.select_columns(x, cols, scale=sc, offset=of)
        rp, col, val = red.to_csr()
        np.testing.assert_array_equal(rp, ro.rowptr)       # bit-exact structure
        np.testing.assert_array_equal(col, ro.col)
        np.testing.assert_array_equal(val, ro.val)         # same single multiply -> bit-exact


@pytest.mark.parametrize("ncols,mode", [(1, None), (40, "sort"), (40, "flat"), (40, None), (200, None), (256, None),
                                        (700, None), (700, "flat"), (1024, None), (1500, None)])
def test_select_columns_paths_and_block_reuse(ctx, synth_small, setopt, ncols, mode):
    """Row-block binning in LDS (<= 1024 columns, the default), flat kernels with bit-mask ranks (KLR_SUBSELECT=flat),
    counters + in-row sort (more columns, or KLR_SUBSELECT=sort), and the pooled block of a
    released reduced matrix handed to the next sub-select: same CSR as featureSelection.Data + Transform (oracle),
    same column sums through the reduced CSC."""
    x, labels = synth_small
    m = upload(ctx, x, labels)
    if mode:
        setopt("KLR_SUBSELECT", mode)
    rng = np.random.default_rng(ncols)
    for rep in range(3):                                   # released blocks are reused by the next call
        k = max(0, ncols - 1 - 7 * rep)
        cols = np.concatenate([[0], 1 + np.sort(rng.choice(x.dim - 1, size=k, replace=False))]).astype(np.int64)
        red = m.select_columns(cols)
        ro = O.select_columns(x, cols)
        rp, col, val = red.to_csr()
        np.testing.assert_array_equal(rp, ro.rowptr)
        np.testing.assert_array_equal(col, ro.col)
        np.testing.assert_array_equal(val, ro.val)
        theta = rng.normal(size=len(cols)) * 0.05
        cw = O.class_weights(labels)
        g = K.gradient_dense(ctx, red, theta, cw)          # column sums over the reduced CSC
        go = O.gradient(ro, labels, theta, cw)
        np.testing.assert_allclose(g, go, rtol=1e-11, atol=1e-13)
        del red


def test_select_columns_drops_explicit_zeros(ctx, scores):
    x, labels, dense = scores
    d2 = dense.copy(); d2[3, 1] = 0.0; d2[9, 6] = 0.0; d2[0, 0] = 0.0
    xz = O.Csr.from_dense(d2)
    m = upload(ctx, xz, labels)
    red = m.select_columns([0, 1, 2, 7]); ro = O.select_columns(xz, [0, 1, 2, 7])
    rp, col, val = red.to_csr()
    np.testing.assert_array_equal(rp, ro.rowptr)
    np.testing.assert_array_equal(col, ro.col)
    np.testing.assert_array_equal(val, ro.val)


@pytest.mark.parametrize("nnz_theta", [0, 5, 60, 1500])
def test_gradient_and_loss_vs_oracle(ctx, synth_small, nnz_theta):
    x, labels = synth_small
    m = upload(ctx, x, labels)
    cw = O.class_weights(labels)
    np.testing.assert_array_equal(m.class_weights(), cw)
    rng = np.random.default_rng(nnz_theta)
    theta = np.zeros(x.dim)
    theta[0] = 0.3
    if nnz_theta:
        idx = 1 + rng.choice(x.dim - 1, size=nnz_theta, replace=False)
        theta[idx] = rng.normal(size=nnz_theta) * 0.05
    g = K.gradient_dense(ctx, m, theta, cw=cw)
    assert_g_close(g, O.gradient(x, labels, theta, cw=cw))
    l = K.loss(ctx, m, theta, cw=cw, lam=0.01)
    assert abs(l - O.loss(x, labels, theta, cw=cw, lam=0.01)) < LOSS_ATOL
    lp = K.log_pdf(ctx, m, theta)
    lpo = np.array([O.class_log_pdf(x, i, theta, 1) for i in range(0, x.n, 37)])
    np.testing.assert_allclose(lp[::37], lpo, rtol=1e-13, atol=1e-15)


def test_gradient_with_transforms(ctx, synth_small):
    x, labels = synth_small
    m = upload(ctx, x, labels)
    rng = np.random.default_rng(5)
    theta = np.zeros(x.dim); theta[0] = -0.2
    idx = 1 + rng.choice(x.dim - 1, size=30, replace=False)
    theta[idx] = rng.normal(size=30) * 0.1
    scale = rng.uniform(0.5, 2.0, size=x.dim); scale[0] = 1.0
    off = rng.uniform(-0.05, 0.05, size=x.dim); off[0] = 0.0
    cw = (0.7, 1.9)
    for sc, of in ((scale, None), (scale, off)):
        g = K.gradient_dense(ctx, m, theta, cw=cw, scale=sc, offset=of)
        assert_g_close(g, O.gradient(x, labels, theta, cw=cw, scale=sc, offset=of))


def test_f64_values_path(ctx):
    # scoresLr-style dense rows with values that are not f32-representable
    rng = np.random.default_rng(11)
    dense = rng.normal(size=(300, 24)) * 3.0 + 0.1
    x = O.Csr.from_dense(dense)
    labels = (rng.uniform(size=300) < 0.4).astype(np.uint8)
    m = upload(ctx, x, labels)
    cw = O.class_weights(labels)
    theta = rng.normal(size=x.dim) * 0.1
    assert_g_close(K.gradient_dense(ctx, m, theta, cw=cw), O.gradient(x, labels, theta, cw=cw))
    assert abs(K.loss(ctx, m, theta, cw=cw) - O.loss(x, labels, theta, cw=cw)) < LOSS_ATOL


@pytest.mark.parametrize("N", [1, 7, 64, 700, 3000])
def test_select_bit_exact(ctx, N):
    rng = np.random.default_rng(N)
    m = 2500
    g = rng.normal(size=m)
    g[rng.choice(m, size=300, replace=False)] = 0.0          # zero gradients (pass 2 only)
    g[100:140] = 0.5; g[900:910] = -0.5                      # exact ties, mixed sign
    g[17] = np.nan
    b = np.zeros(m + 1, dtype=np.uint8); b[0] = 1
    pre = rng.choice(m, size=min(N // 2, 20), replace=False)
    b[pre + 1] = 1
    ok, lam, c, b2 = K.select(ctx, g, b, len(pre), N, lambda_prev=0.3, eps_lambda=0.0)
    oko, lamo, co, bo = O.select(np.nan_to_num(g, nan=0.0), b, len(pre), N, lambda_prev=0.3, eps_lambda=0.0)
    assert (ok, c) == (oko, co)
    assert lam == lamo                                       # bit-exact
    np.testing.assert_array_equal(b2, bo)


@pytest.mark.parametrize("m,N,kind", [(300000, 100, "normal"), (1048576, 500, "normal"), (400000, 50, "clustered"),
                                      (65536, 100, "clustered"), (500000, 2500, "lognormal")])
def test_select_large_m_bit_exact(ctx, m, N, kind):
    # m > 262144 takes the multi-workgroup selection kernel; "clustered" puts thousands of keys into the
    # bracketing coarse bin (second-level trim / bitonic / exact fallback paths)
    rng = np.random.default_rng(m + N)
    if kind == "normal":
        g = rng.normal(size=m)
    elif kind == "lognormal":
        g = rng.lognormal(sigma=4.0, size=m) * rng.choice([-1.0, 1.0], size=m)
    else:
        g = 1.0 + 1e-6 * rng.random(size=m)                   # all within one quarter-octave
        g[rng.choice(m, size=m // 2, replace=False)] *= -1.0
    g[rng.choice(m, size=m // 10, replace=False)] = 0.0
    b = np.zeros(m + 1, dtype=np.uint8); b[0] = 1
    pre = rng.choice(m, size=10, replace=False)
    b[pre + 1] = 1
    ok, lam, c, b2 = K.select(ctx, g, b, len(pre), N, lambda_prev=0.3, eps_lambda=0.0)
    oko, lamo, co, bo = O.select(g, b, len(pre), N, lambda_prev=0.3, eps_lambda=0.0)
    assert (ok, c) == (oko, co)
    assert lam == lamo
    np.testing.assert_array_equal(b2, bo)


def test_select_ties_across_threshold_and_all_zero(ctx):
    # the 2N-th largest sits inside a block of ties: lowest column ids must be taken
    g = np.zeros(1000); g[200:260] = 1.0; g[5] = 2.0
    b = np.zeros(1001, dtype=np.uint8); b[0] = 1
    for N in (3, 10, 31):
        r = K.select(ctx, g, b, 0, N); ro = O.select(g, b, 0, N)
        assert r[:3] == ro[:3]
        np.testing.assert_array_equal(r[3], ro[3])
    g0 = np.zeros(50)
    r = K.select(ctx, g0, np.r_[1, np.zeros(50)].astype(np.uint8), 0, 4)
    ro = O.select(g0, np.r_[1, np.zeros(50)].astype(np.uint8), 0, 4)
    assert r[:3] == ro[:3]
    np.testing.assert_array_equal(r[3], ro[3])


def test_select_fixed(ctx, scores):
    x, labels, _ = scores
    g = O.gradient(x, labels, np.zeros(x.dim))
    b = np.zeros(x.dim, dtype=np.uint8); b[0] = 1
    for lam, mf in ((1.0, 0), (1.0, 1), (0.05, 3)):
        r = K.select_fixed(ctx, g, b, 0, lam, mf); ro = O.select_fixed(g, b, 0, lam, mf)
        assert r[:2] == ro[:2]
        np.testing.assert_array_equal(r[2], ro[2])


def leapfrog_epoch_oracle(x, labels, cw, cols, theta_red, N, lam_prev):
    theta = np.zeros(x.dim); theta[cols] = theta_red
    g = O.gradient(x, labels, theta, cw=cw)
    b = np.zeros(x.dim, dtype=np.uint8); b[0] = 1
    nzc = [c for c, t in zip(cols[1:], theta_red[1:]) if t != 0.0]
    b[nzc] = 1
    ok, lam, c, b2 = O.select(g[1:], b, len(nzc), N, lambda_prev=lam_prev)
    return ok, lam, c, b2, O.loss(x, labels, theta, cw=cw), g


def test_sweep_select_and_cd_solve_leapfrog(ctx, synth_small):
    """Two leapfrog epochs end to end: identical selections, lambda to rounding, theta to 1e-6."""
    x, labels = synth_small
    m = upload(ctx, x, labels)
    cw = O.class_weights(labels)
    N = 12
    # epoch 0: theta = 0
    r = K.sweep_select(ctx, m, None, None, np.zeros(1), cw, N, want_g=True)
    ok, lam, c, b2, l0, g = leapfrog_epoch_oracle(x, labels, cw, np.array([0]), np.zeros(1), N, 0.0)
    assert (r["ok"], r["c"]) == (ok, c)
    np.testing.assert_array_equal(r["b"], b2)
    assert abs(r["lam"] - lam) <= 1e-12 * lam
    assert abs(r["loss"] - l0) < LOSS_ATOL
    assert_g_close(r["g"], g)
    cols = np.nonzero(b2)[0]
    red = m.select_columns(cols); redo = O.select_columns(x, cols)
    th_gpu, recs = K.cd_solve(ctx, red, np.zeros(len(cols)), cw=cw, l1reg=r["lam"] * x.n, eps_loss=1e-8)
    th_ora, reco = O.cd_solve(redo, labels, np.zeros(len(cols)), cw=cw, l1reg=lam * x.n, eps_loss=1e-8)
    assert len(recs) == len(reco)
    assert [a["iteration"] for a in recs] == [a["iteration"] for a in reco]
    assert [a["nonzero"] for a in recs] == [a["nonzero"] for a in reco]
    np.testing.assert_allclose([a["loss"] for a in recs], [a["loss"] for a in reco], rtol=0, atol=1e-11)
    np.testing.assert_allclose(th_gpu, th_ora, rtol=THETA_RTOL, atol=1e-12)
    # epoch 1 from the solved theta
    r1 = K.sweep_select(ctx, m, red, cols, th_gpu, cw, N, lambda_prev=r["lam"] * x.n, want_g=True)
    ok1, lam1, c1, b1, l1, g1 = leapfrog_epoch_oracle(x, labels, cw, cols, th_ora, N, lam * x.n)
    assert (r1["ok"], r1["c"]) == (ok1, c1)
    np.testing.assert_array_equal(r1["b"], b1)
    assert abs(r1["lam"] - lam1) <= 1e-6 * lam1
    assert abs(r1["loss"] - l1) < 1e-9


def test_cd_solve_tight_matches_oracle_and_kkt(ctx, scores):
    x, labels, _ = scores
    m = upload(ctx, x, labels)
    red = m.select_columns([0, 2, 7]); redo = O.select_columns(x, [0, 2, 7])
    lam = 2.496875
    th, recs = K.cd_solve(ctx, red, np.zeros(3), l1reg=lam * 16, eps=1e-13, eps_loss=0.0, max_iter=100000)
    np.testing.assert_allclose(th, [0.851127187641263, -0.05490113203904316, -0.03053978552591902], rtol=1e-8)
    tho, _ = O.cd_solve(redo, labels, np.zeros(3), l1reg=lam * 16, eps=1e-13, eps_loss=0.0, max_iter=100000)
    np.testing.assert_allclose(th, tho, rtol=1e-9)
    # max_iter cap is honoured identically
    th5, r5 = K.cd_solve(ctx, red, np.zeros(3), l1reg=lam * 16, eps_loss=1e-8, max_iter=5)
    to5, o5 = O.cd_solve(redo, labels, np.zeros(3), l1reg=lam * 16, eps_loss=1e-8, max_iter=5)
    assert len(r5) == len(o5)
    np.testing.assert_allclose(th5, to5, rtol=1e-9)


def test_edge_cases(ctx):
    # a matrix whose rows hold only the bias; a single row; ragged rows
    rowptr = np.array([0, 1, 2, 3], dtype=np.int64); col = np.zeros(3, dtype=np.int64); val = np.ones(3)
    x = O.Csr(3, 5, rowptr, col, val); labels = np.array([1, 0, 1], dtype=np.uint8)
    m = upload(ctx, x, labels)
    theta = np.array([0.4, 0, 0, 0, 0.0])
    assert_g_close(K.gradient_dense(ctx, m, theta), O.gradient(x, labels, theta))
    assert abs(K.loss(ctx, m, theta) - O.loss(x, labels, theta)) < LOSS_ATOL
    rowptr = np.array([0, 1, 6, 8], dtype=np.int64)
    col = np.array([0, 0, 1, 2, 3, 4, 0, 4], dtype=np.int64); val = np.array([1, 1, 2, 3, 4, 5, 1, 7.0])
    x = O.Csr(3, 5, rowptr, col, val)
    m = upload(ctx, x, labels)
    theta = np.array([0.1, -0.2, 0.3, 0.0, 0.05])
    assert_g_close(K.gradient_dense(ctx, m, theta), O.gradient(x, labels, theta))
    red = m.select_columns([0, 3, 4]); ro = O.select_columns(x, [0, 3, 4])
    rp, c, v = red.to_csr()
    np.testing.assert_array_equal(rp, ro.rowptr); np.testing.assert_array_equal(c, ro.col)
    # invalid input is refused like the reference's panic("internal error")
    with pytest.raises(K.KlrError):
        K.Matrix.from_csr(ctx, 1, 3, np.array([0, 1]), np.array([1]), np.array([1.0]), np.array([1], dtype=np.uint8))
    with pytest.raises(K.KlrError):
        m.select_columns([1, 2])


def test_sharded_partials_sum_to_whole(ctx, synth_small):
    """Row shards with n_total: partial gradients / losses add up to the unsharded result."""
    x, labels = synth_small
    cw = O.class_weights(labels)
    theta = np.zeros(x.dim); theta[0] = 0.1; theta[[5, 77, 900]] = [0.2, -0.1, 0.05]
    g_ref = O.gradient(x, labels, theta, cw=cw)
    g_sum = np.zeros(x.dim); l_sum = 0.0
    for (r0, r1) in ((0, 250), (250, 600)):
        rp = x.rowptr[r0:r1 + 1] - x.rowptr[r0]
        sl = slice(x.rowptr[r0], x.rowptr[r1])
        xs = O.Csr(r1 - r0, x.dim, rp, x.col[sl], x.val[sl])
        ms = upload(ctx, xs, labels[r0:r1], n_total=x.n)
        g_sum += K.gradient_dense(ctx, ms, theta, cw=cw)
        l_sum += K.loss(ctx, ms, theta, cw=cw)      # each shard already divides by n_total
    assert_g_close(g_sum, g_ref)
    assert abs(l_sum - O.loss(x, labels, theta, cw=cw)) < LOSS_ATOL


# ---------------------------------------------------------------------------------------------------
# the row-tiled packed layout (large count matrices) -- the roofline kernel's own parity cases
# ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def synth_tiled():
    # many row tiles (the geometry gives one tile per workgroup here), ragged last tile, ~1.6e6 nonzeros
    x, labels = O.synth_csr(20261004, 0, 20000, 20000, 3000, 80, 24)
    return x, labels


@pytest.fixture(params=["q64c12", "q64c13", "r64", "u24", "u32"])
def entry_format(request, setopt):
    # the packings of the tiled stream: 8-byte quads (row | 4 columns) / 3-byte unit entries / 4-byte entries with a count
    setopt("KLR_ENTRY_FORMAT", request.param)
    return request.param


def test_tiled_sweep_matches_oracle(ctx, synth_tiled, entry_format):
    x, labels = synth_tiled
    m = upload(ctx, x, labels)
    lay = m.layout()
    assert lay["tiled"] and lay["format"] == entry_format
    if entry_format.startswith("q64"):
        assert lay["tile_rows"] == x.n                   # row-streamed: strips only, no row tiles
    else:
        assert 0 < lay["tile_rows"] <= 4094 and lay["tile_rows"] % 2 == 0
    assert lay["value_bytes"] == {"q64c12": -1, "q64c13": -1, "r64": -1, "u24": 0, "u32": 1}[entry_format]
    cw = O.class_weights(labels)
    rng = np.random.default_rng(1)
    for k in (0, 40):
        theta = np.zeros(x.dim); theta[0] = -0.1
        if k:
            idx = 1 + rng.choice(x.dim - 1, size=k, replace=False)
            theta[idx] = rng.normal(size=k) * 0.1
        g = K.gradient_dense(ctx, m, theta, cw=cw)
        assert_g_close(g, O.gradient(x, labels, theta, cw=cw))
    # run-to-run bitwise reproducibility of the device path
    g2 = K.gradient_dense(ctx, m, theta, cw=cw)
    np.testing.assert_array_equal(g, g2)


@pytest.mark.parametrize("fmt,dim", [("q64c12", 4_200_000), ("q64c13", 8_300_000)])
def test_q64_many_strips_per_workgroup(ctx, setopt, fmt, dim):
    # > 1000 column strips on 256 workgroups: every workgroup streams four or five pieces, so the second accumulator
    # region is written out and reused in mid-stream (a piece boundary with and without a barrier), and most strips hold
    # only a block or two.  g at theta = 0 is an exact function of the integer column sums: bitwise against numpy.
    setopt("KLR_ENTRY_FORMAT", fmt)
    rng = np.random.default_rng(11)
    n, per_row = 6000, 400
    cols = np.sort(rng.integers(1, dim, size=(n, per_row)), axis=1)
    keep = np.ones_like(cols, dtype=bool); keep[:, 1:] = cols[:, 1:] != cols[:, :-1]
    counts = rng.integers(1, 4, size=(n, per_row)).astype(np.float64)
    rowlen = 1 + keep.sum(axis=1)
    rowptr = np.zeros(n + 1, dtype=np.int64); np.cumsum(rowlen, out=rowptr[1:])
    col = np.zeros(rowptr[-1], dtype=np.int64); val = np.ones(rowptr[-1])
    pos = (rowptr[:-1, None] + 1 + np.cumsum(keep, axis=1) - 1)[keep]
    col[pos] = cols[keep]; val[pos] = counts[keep]
    labels = (rng.uniform(size=n) < 0.4).astype(np.uint8)
    m = K.Matrix.from_csr(ctx, n, dim, rowptr, col, val, labels)
    lay = m.layout()
    assert lay["format"] == fmt and lay["nseg"] > 1000
    cw = m.class_weights()
    out = K.sweep_select(ctx, m, None, None, np.zeros(1), cw, 50, want_g=True)
    # s_i = cw_i (0.5 - y_i) / n at theta = 0 -> exact fixed-point sums; compare with the f64 column sums within 1e-11 max|g|,
    # and bitwise with a second build in the other order of the same sums (the CSC path)
    r = (0.5 - labels.astype(np.float64)) * np.where(labels > 0, cw[1], cw[0]) / n
    g_ref = np.zeros(dim); np.add.at(g_ref, col, val * np.repeat(r, rowlen))
    assert np.max(np.abs(out["g"] - g_ref)) <= 1e-11 * np.max(np.abs(g_ref))
    out2 = K.sweep_select(ctx, m, None, None, np.zeros(1), cw, 50, want_g=True)
    np.testing.assert_array_equal(out["g"], out2["g"])
    setopt("KLR_ENTRY_FORMAT", "u32")
    m2 = K.Matrix.from_csr(ctx, n, dim, rowptr, col, val, labels)
    out3 = K.sweep_select(ctx, m2, None, None, np.zeros(1), cw, 50, want_g=True)
    np.testing.assert_array_equal(out["g"], out3["g"])
    np.testing.assert_array_equal(out["b"], out3["b"])
    assert out["lam"] == out3["lam"] and out["c"] == out3["c"] == 50
    m.close(); m2.close()


def test_tiled_dense_counts(ctx, entry_format):
    # dense count data: every (row, column) pair present, many same-column LDS atomics per wave
    rng = np.random.default_rng(2)
    dense = rng.integers(0, 5, size=(9000, 130)).astype(np.float64)
    x = O.Csr.from_dense(dense)
    labels = (rng.uniform(size=9000) < 0.5).astype(np.uint8)
    m = upload(ctx, x, labels)
    assert m.layout()["tiled"]
    theta = rng.normal(size=x.dim) * 0.01
    assert_g_close(K.gradient_dense(ctx, m, theta), O.gradient(x, labels, theta))


def test_tiled_large_counts_and_unpackable_values(ctx, entry_format):
    rng = np.random.default_rng(3)
    x, labels = O.synth_csr(5, 0, 9000, 9000, 500, 150, 4)
    v = x.val.copy()
    v[1::7] = 70000.0                       # counts above 255 are split into several entries
    v[2::13] = 255.0; v[5::17] = 256.0
    v[x.rowptr[:-1]] = 1.0                  # keep the bias entries
    x1 = O.Csr(x.n, x.dim, x.rowptr, x.col, v)
    m1 = upload(ctx, x1, labels)
    assert m1.layout()["tiled"]
    theta = np.zeros(x.dim); theta[0] = 0.2
    assert_g_close(K.gradient_dense(ctx, m1, theta), O.gradient(x1, labels, theta))
    v[3::11] = 2.5                          # not a count: the plain CSC path must take over
    v[x.rowptr[:-1]] = 1.0
    x2 = O.Csr(x.n, x.dim, x.rowptr, x.col, v)
    m2 = upload(ctx, x2, labels)
    assert not m2.layout()["tiled"]
    assert_g_close(K.gradient_dense(ctx, m2, theta), O.gradient(x2, labels, theta))


def test_full_size_properties(ctx):
    """BASELINE configs[2] size (100k x 65,536, ~1e8 nonzeros): size-independent properties.
    At theta = 0, cw = [1,1]:  g_j = (bg column sum - fg column sum) / (2n), exactly computable
    from integer column sums; plus selection sortedness and sweep idempotence."""
    n, m, draws, N = 100000, 65536, 993, 100
    rowptr, col, val, labels = K.synth_csr(20261004, 0, n, n, m, draws, 2 * N)
    mat = K.Matrix.from_csr(ctx, n, m + 1, rowptr, col, val, labels)
    assert mat.layout()["tiled"] and mat.layout()["value_bytes"] == -1     # k-mer counts: 8-byte quads, 2 bytes per unit entry
    rows_fg = int(rowptr[n // 2])
    fg = np.bincount(col[:rows_fg], weights=val[:rows_fg], minlength=m + 1)
    bg = np.bincount(col[rows_fg:], weights=val[rows_fg:], minlength=m + 1)
    g_exact = (bg - fg) / (2.0 * n)
    r = K.sweep_select(ctx, mat, None, None, np.zeros(1), (1.0, 1.0), N, want_g=True)
    assert np.max(np.abs(r["g"] - g_exact)) <= 1e-12 * np.max(np.abs(g_exact))
    assert abs(r["loss"] - np.log(2.0)) < 1e-12
    # the selected set is exactly the N largest |g| (ties impossible to straddle: checked)
    order = np.lexsort((np.arange(m), -np.abs(g_exact[1:])))
    assert abs(g_exact[1:][order[N - 1]]) > abs(g_exact[1:][order[N]])
    assert sorted(np.nonzero(r["b"])[0][1:] - 1) == sorted(order[:N])
    v, w = abs(g_exact[1:][order[N - 1]]), np.max(np.abs(g_exact[1:])[np.abs(g_exact[1:]) < abs(g_exact[1:][order[N - 1]])])
    assert abs(r["lam"] - (v + w) / 2) <= 1e-12 * r["lam"]
    # idempotence / reproducibility: the same sweep twice gives bit-identical output
    r2 = K.sweep_select(ctx, mat, None, None, np.zeros(1), (1.0, 1.0), N, want_g=True)
    np.testing.assert_array_equal(r["g"], r2["g"])
    assert r["lam"] == r2["lam"]
    # linearity in the class weights: g(cw=[a,b]) = a*g_bg_part + b*g_fg_part
    ra = K.sweep_select(ctx, mat, None, None, np.zeros(1), (2.0, 0.5), N, want_g=True)
    g_lin = (2.0 * bg - 0.5 * fg) / (2.0 * n)
    assert np.max(np.abs(ra["g"] - g_lin)) <= 1e-12 * np.max(np.abs(g_lin))


@pytest.mark.parametrize("gram_path", [None, "lds", "global"])
@pytest.mark.parametrize("N", [40, 150, 300, 560, 1100, 2100])
def test_cd_solve_solver_variants(ctx, N, gram_path, setopt):
    """d = N+1 selects the solver kernel: one wavefront (Gram in LDS, or 4/8 q values per lane with rows from L2),
    blocked Gauss-Seidel over 16 wavefronts (d > 512), super-blocks with a multi-workgroup update (d > 2048);
    all must follow the oracle's iterate sequence.  gram_path forces the Gram build: privatised in LDS
    (d <= 128: whole, d <= 1024: tile pairs) or global integer atomics -- the default picks by the number of pairs."""
    if gram_path:
        setopt("KLR_GRAM_PATH", gram_path)
    x, labels = O.synth_csr(77, 0, 3000, 3000, 8192, 400, 2 * N)
    m = upload(ctx, x, labels)
    cw = O.class_weights(labels)
    r = K.sweep_select(ctx, m, None, None, np.zeros(1), cw, N)
    cols = np.nonzero(r["b"])[0]
    assert len(cols) == N + 1
    red = m.select_columns(cols); redo = O.select_columns(x, cols)
    th, recs = K.cd_solve(ctx, red, np.zeros(N + 1), cw=cw, l1reg=r["lam"] * x.n, eps_loss=1e-9, max_iter=60)
    tho, reco = O.cd_solve(redo, labels, np.zeros(N + 1), cw=cw, l1reg=r["lam"] * x.n, eps_loss=1e-9, max_iter=60)
    assert [a["iteration"] for a in recs] == [a["iteration"] for a in reco]
    assert [a["nonzero"] for a in recs] == [a["nonzero"] for a in reco]
    np.testing.assert_allclose([a["loss"] for a in recs], [a["loss"] for a in reco], rtol=0, atol=1e-10)
    np.testing.assert_allclose(th, tho, rtol=THETA_RTOL, atol=1e-11)


@pytest.mark.parametrize("kind", ["standardizer", "variance-scaler", "max-abs-scaler", "mean-scaler"])
def test_transform_fit_on_device(ctx, synth_small, kind):
    x, labels = synth_small
    m = upload(ctx, x, labels)
    sc, of = K.transform_fit(ctx, m, kind)
    sco, ofo = O.transform_fit(x, kind)
    ok = np.isfinite(sco)
    np.testing.assert_allclose(sc[ok], sco[ok], rtol=1e-10)
    np.testing.assert_array_equal(np.isfinite(sc), ok)               # empty columns: 1/0 = inf on both sides
    assert (of is None) == (ofo is None)
    if of is not None:
        np.testing.assert_allclose(of, ofo, rtol=1e-12, atol=1e-15)


# ---------------------------------------------------------------------------------------------------
# co-occurrence features: the pair matrix written out on the device
# ---------------------------------------------------------------------------------------------------
def _assert_same_csr(m, e):
    rp, col, val = m.to_csr()
    np.testing.assert_array_equal(rp, e.rowptr)
    np.testing.assert_array_equal(col, e.col)
    np.testing.assert_array_equal(val, e.val)


def test_cooccurrence_expansion_counts(ctx):
    # k-mer counts restricted to 40 pre-selected columns: products of counts stay count data (device ingest)
    x, labels = O.synth_csr(7, 0, 3000, 3000, 400, 120, 8)
    cols = np.r_[0, 1 + np.sort(np.random.default_rng(5).choice(400, size=40, replace=False))]
    xs = O.select_columns(x, cols)
    base = upload(ctx, x, labels).select_columns(cols)
    m = base.expand_cooccurrence()
    e = O.expand_cooccurrence(xs)
    assert m.dim == e.dim == 40 * 41 // 2 + 1
    _assert_same_csr(m, e)
    theta = np.zeros(e.dim); theta[0] = 0.1
    theta[np.random.default_rng(6).choice(e.dim - 1, size=30, replace=False) + 1] = 0.01
    assert_g_close(K.gradient_dense(ctx, m, theta), O.gradient(e, labels, theta))
    assert abs(K.loss(ctx, m, theta) - O.loss(e, labels, theta)) < LOSS_ATOL


def test_cooccurrence_expansion_scores_and_golden_selection(ctx, golden_dir):
    # TestScores2 (scoresLr_test.go:64-91): real-valued scores; --lambda-auto=1 --co-occurrence picks the pair (1, 2)
    import os
    _, fg = O.read_scores_table(os.path.join(golden_dir, "scoresLr_test_co_fg.table"))
    _, bg = O.read_scores_table(os.path.join(golden_dir, "scoresLr_test_co_bg.table"))
    dense = np.concatenate([fg, bg], axis=0)
    labels = np.array([1] * len(fg) + [0] * len(bg), dtype=np.uint8)
    x = O.Csr.from_dense(dense)
    m = upload(ctx, x, labels).expand_cooccurrence()
    e = O.expand_cooccurrence(x)
    _assert_same_csr(m, e)
    r = K.sweep_select(ctx, m, None, None, np.zeros(1), (1.0, 1.0), 1, want_g=True)
    assert_g_close(r["g"], O.gradient(e, labels, np.zeros(e.dim)))
    picked = np.nonzero(r["b"])[0][1:]
    assert len(picked) == 1 and K.coeff_sub2ind(dense.shape[1], int(picked[0]) - 1) == (1, 2)


def test_cooccurrence_inherits_row_mask(ctx):
    x, labels = O.synth_csr(9, 0, 2500, 2500, 60, 40, 4)
    mat = upload(ctx, x, labels)
    mask = (np.arange(x.n) % 5 != 2).astype(np.uint8)
    mat.set_active_rows(mask)
    m = mat.expand_cooccurrence()
    keep = np.nonzero(mask)[0]
    xe = O.expand_cooccurrence(x)
    sub = O.Csr.from_rows(xe, keep) if hasattr(O.Csr, "from_rows") else None
    theta = np.zeros(xe.dim); theta[0] = -0.2; theta[5] = 0.03
    g = K.gradient_dense(ctx, m, theta)
    if sub is not None:
        assert_g_close(g, O.gradient(sub, labels[keep], theta))
    else:
        # the masked gradient equals the gradient of the kept rows: compare through a re-upload of those rows
        d = xe.to_dense()[keep]
        assert_g_close(g, O.gradient(O.Csr.from_dense(d[:, 1:]), labels[keep], theta))


def test_hinted_selection_sequence_matches_oracle(ctx, synth_tiled):
    """Consecutive sweeps on one context reuse the previous K-th |g| as a compaction hint (SelHint): the selection
    must stay exact when the hint fits, when the gradient collapses under it (too few candidates -> regular path),
    when it is far too low, and when N changes."""
    x, labels = synth_tiled
    m = upload(ctx, x, labels)
    cw = O.class_weights(labels)
    rng = np.random.default_rng(11)
    thetas = []
    t = np.zeros(x.dim); t[0] = -0.1; thetas.append((t.copy(), 20))
    t[1 + rng.choice(x.dim - 1, size=30, replace=False)] = rng.normal(size=30) * 0.05; thetas.append((t.copy(), 20))
    thetas.append((t.copy(), 20))                                   # identical sweep: hint fits exactly
    big = t.copy(); big[0] = 30.0; thetas.append((big, 20))          # saturated sigmoid: gradients collapse, hint too high
    thetas.append((t.copy(), 20))                                   # back up: the tiny hint now admits nearly everything
    thetas.append((t.copy(), 7)); thetas.append((t.copy(), 300))     # N changes: hint off, then re-armed
    for th, N in thetas:
        idx = np.nonzero(th)[0]
        red = m.select_columns(idx if idx[0] == 0 else np.r_[0, idx])
        cols = idx if idx[0] == 0 else np.r_[0, idx]
        r = K.sweep_select(ctx, m, red, cols, th[cols], cw, N, want_g=True)
        g = O.gradient(x, labels, th, cw=cw)
        assert_g_close(r["g"], g)
        b = np.zeros(x.dim, dtype=np.uint8); b[0] = 1
        b[cols[1:][th[cols[1:]] != 0.0]] = 1
        c_prev = int(b.sum()) - 1
        # selection on OUR gradient (bitwise what the device saw) through the oracle's rule
        ok, lam, c, b2 = O.select(r["g"][1:], b, c_prev, N)
        assert (r["ok"], r["c"]) == (ok, c)
        assert r["lam"] == lam
        np.testing.assert_array_equal(r["b"], b2)


def test_tiled_ragged_rows_and_empty_tiles(ctx, entry_format):
    """Tiled stream corner cases: bias-only rows, a column present in every row, whole column tiles without entries,
    counts that need several quads / split entries, a last row tile with a handful of rows."""
    rng = np.random.default_rng(21)
    n, m = 8200, 9000                                   # three column tiles, many row tiles, ragged last one
    rows, cols, vals = [], [], []
    for i in range(n):
        if i % 17 == 0:
            k = np.array([], dtype=np.int64)            # bias-only row
        else:
            k = np.unique(rng.integers(1, 3000, size=int(rng.integers(1, 400))))      # only the first column tile is used...
            if i % 5 == 0:
                k = np.unique(np.r_[k, 8999])           # ...except one far column; tile 2 stays empty
            k = np.unique(np.r_[k, 7])                  # column 7 in every non-empty row
        v = rng.integers(1, 4, size=len(k)).astype(np.float64)
        if len(k) and i % 11 == 0:
            v[0] = 300.0                                # a count above 255
        rows.append(len(k)); cols.append(k); vals.append(v)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    col = []; val = []
    for i in range(n):
        col.append(np.r_[0, cols[i]]); val.append(np.r_[1.0, vals[i]])
        rowptr[i + 1] = rowptr[i] + 1 + rows[i]
    x = O.Csr(n, m + 1, rowptr, np.concatenate(col).astype(np.int64), np.concatenate(val))
    assert x.nnz >= (1 << 20)
    labels = (rng.uniform(size=n) < 0.4).astype(np.uint8)
    mat = upload(ctx, x, labels)
    assert mat.layout()["tiled"]
    theta = np.zeros(x.dim); theta[0] = 0.3; theta[7] = -0.02; theta[8999] = 0.5
    assert_g_close(K.gradient_dense(ctx, mat, theta), O.gradient(x, labels, theta))
    r = K.sweep_select(ctx, mat, None, None, np.zeros(1), (1.0, 1.0), 5, want_g=True)
    assert_g_close(r["g"], O.gradient(x, labels, np.zeros(x.dim)))
    rp, cc, vv = mat.to_csr()
    np.testing.assert_array_equal(rp, x.rowptr); np.testing.assert_array_equal(cc, x.col); np.testing.assert_array_equal(vv, x.val)


@pytest.mark.parametrize("n,m", [(4000, 60), (3000, 255), (700, 17)])
def test_dense_panel_mfma_gram(ctx, n, m, setopt):
    """Dense score tables (every column stored, scoresLr_data.go:108-117): the reduced matrix is a dense n x d panel and
    the Gram matrix of a re-weighting is built on the matrix cores (k_gram_dense_mfma, v_mfma_f64_16x16x4_f64).  The
    solve must follow the oracle's iterate sequence (theta within 1e-6) and agree with the sparse Gram builds."""
    rng = np.random.default_rng(12)
    dense = np.c_[np.ones(n), rng.normal(size=(n, m))]
    beta = np.zeros(m + 1); beta[1:9] = rng.normal(size=8)
    labels = (rng.uniform(size=n) < 1.0 / (1.0 + np.exp(-dense @ beta))).astype(np.uint8)
    rowptr = np.arange(n + 1, dtype=np.int64) * (m + 1)
    col = np.tile(np.arange(m + 1, dtype=np.int64), n)
    x = O.Csr(n, m + 1, rowptr, col, dense.reshape(-1).copy())
    mat = upload(ctx, x, labels)
    assert not mat.layout()["tiled"]
    cw = O.class_weights(labels)
    cols = np.arange(m + 1)
    red = mat.select_columns(cols)
    assert red.nnz == n * (m + 1)
    l1 = 0.02 * n
    th, recs = K.cd_solve(ctx, red, np.zeros(m + 1), cw=cw, l1reg=l1, eps_loss=1e-10, max_iter=80)
    tho, reco = O.cd_solve(x, labels, np.zeros(m + 1), cw=cw, l1reg=l1, eps_loss=1e-10, max_iter=80)
    assert [a["iteration"] for a in recs] == [a["iteration"] for a in reco]
    assert [a["nonzero"] for a in recs] == [a["nonzero"] for a in reco]
    np.testing.assert_allclose(th, tho, rtol=THETA_RTOL, atol=1e-10)
    setopt("KLR_GRAM_NO_MFMA", "1")
    th2, recs2 = K.cd_solve(ctx, red, np.zeros(m + 1), cw=cw, l1reg=l1, eps_loss=1e-10, max_iter=80)
    assert len(recs2) == len(recs)
    np.testing.assert_allclose(th, th2, rtol=1e-8, atol=1e-12)


@pytest.mark.parametrize("n,dim,per_row,fmt", [
    (3700, 9000, 300, "q64c12"),        # 3 strips, ~10 rows per block: every workgroup's rows fit LDS (S copied there)
    (3700, 9000, 300, "q64c13"),
    (3000, 40000, 3000, "q64c13"),      # 5 wide strips, dense rows: the LDS copy beside two 64 KiB regions
    (40000, 5000, 60, "q64c12"),        # 2 strips, 20k rows per strip: too many rows per workgroup for the LDS copy
    (2100, 600000, 520, "q64c12"),      # 149 sparse strips of ~16 blocks: more strips than half the workgroups, short pieces
    (9000, 4033, 500, "q64c12"),        # the second strip holds a single column
])
def test_q64_shapes_bitwise_against_u32(ctx, setopt, n, dim, per_row, fmt):
    """The row-streamed kernel in its variants (S from global memory or copied to LDS, one or two pieces per workgroup,
    strips of a few blocks) against the tiled u32 format: the same integers, bit for bit, and the f64 column sums."""
    rng = np.random.default_rng(n + dim)
    cols = np.sort(rng.integers(1, dim, size=(n, per_row)), axis=1)
    keep = np.ones_like(cols, dtype=bool); keep[:, 1:] = cols[:, 1:] != cols[:, :-1]
    counts = rng.integers(1, 5, size=(n, per_row)).astype(np.float64)
    rowlen = 1 + keep.sum(axis=1)
    rowptr = np.zeros(n + 1, dtype=np.int64); np.cumsum(rowlen, out=rowptr[1:])
    col = np.zeros(rowptr[-1], dtype=np.int64); val = np.ones(rowptr[-1])
    pos = (rowptr[:-1, None] + np.cumsum(keep, axis=1))[keep]
    col[pos] = cols[keep]; val[pos] = counts[keep]
    labels = (rng.uniform(size=n) < 0.5).astype(np.uint8)
    if rowptr[-1] < (1 << 20):
        pytest.skip("below the packed stream's size threshold")
    setopt("KLR_ENTRY_FORMAT", fmt)
    m = K.Matrix.from_csr(ctx, n, dim, rowptr, col, val, labels)
    assert m.layout()["format"] == fmt
    cw = m.class_weights()
    setopt("KLR_ENTRY_FORMAT", "u32")
    m2 = K.Matrix.from_csr(ctx, n, dim, rowptr, col, val, labels)
    assert m2.layout()["format"] == "u32"
    theta = np.zeros(dim); theta[0] = 0.3
    idx = 1 + rng.choice(dim - 1, size=25, replace=False); theta[idx] = rng.normal(size=25) * 0.05
    for th in (np.zeros(dim), theta):
        g1 = K.gradient_dense(ctx, m, th, cw=cw)
        g2 = K.gradient_dense(ctx, m2, th, cw=cw)
        np.testing.assert_array_equal(g1, g2)
    r = (0.5 - labels.astype(np.float64)) * np.where(labels > 0, cw[1], cw[0]) / n
    g_ref = np.zeros(dim); np.add.at(g_ref, col, val * np.repeat(r, rowlen))
    g0 = K.gradient_dense(ctx, m, np.zeros(dim), cw=cw)
    assert np.max(np.abs(g0 - g_ref)) <= 1e-11 * np.max(np.abs(g_ref))
    m.close(); m2.close()


@pytest.mark.parametrize("switch", ["KLR_Q64_SLDS=0", "KLR_Q64_WGS=2", "KLR_Q64_BANK=0", "KLR_FINISH_NO_FLAT=1", "KLR_Q64_CUT=snap",
                                    "KLR_Q64_PIECE_COST=0", "KLR_Q64_SNAP_TOL=0", "KLR_Q64_LDS_KIB=0"])
def test_q64_switches_leave_the_sums_alone(ctx, synth_tiled, setopt, switch):
    """Every ingest / launch switch of the row-streamed sweep (column order, cuts, S in LDS, regions per workgroup,
    finish form) changes the schedule only: the gradient is the same, bit for bit."""
    x, labels = synth_tiled
    setopt("KLR_ENTRY_FORMAT", "q64c12")
    m0 = upload(ctx, x, labels)
    cw = O.class_weights(labels)
    rng = np.random.default_rng(4)
    theta = np.zeros(x.dim); theta[0] = 0.1
    theta[1 + rng.choice(x.dim - 1, size=30, replace=False)] = rng.normal(size=30) * 0.1
    g0 = K.gradient_dense(ctx, m0, theta, cw=cw)
    name, value = switch.split("=")
    setopt(name, value)
    m1 = upload(ctx, x, labels)
    assert m1.layout()["format"] == "q64c12"
    np.testing.assert_array_equal(K.gradient_dense(ctx, m1, theta, cw=cw), g0)
    assert_g_close(g0, O.gradient(x, labels, theta, cw=cw))
