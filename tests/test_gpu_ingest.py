"""GPU: device ingest (FASTA sequences -> k-mer counts -> CSR / CSC / tiled stream, all on the device) against the
host ingest of the same counts.  Integer work: everything must be bit-identical."""
import os

import numpy as np
import pytest

import kmerlr_amd as K
from oracle import oracle as O
from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = K.Context(0)
    yield c
    c.close()


def read_fasta(path):
    seqs, cur = [], None
    for line in open(path):
        line = line.strip()
        if line.startswith(">"):
            if cur is not None:
                seqs.append(cur)
            cur = ""
        elif cur is not None:
            cur += line
    if cur is not None:
        seqs.append(cur)
    return seqs


def host_counts(seqs, M, N, revcomp=False, binarize=False):
    """Reference counting in Python: column ids for k = M..N in turn, classes in increasing base-4 code order."""
    code = {"a": 0, "c": 1, "g": 2, "t": 3}

    def rc(c, k):
        r = 0
        for _ in range(k):
            r = (r << 2) | (3 - (c & 3)); c >>= 2
        return r
    maps, base = [], 0
    for k in range(M, N + 1):
        if revcomp:
            reps = sorted({min(c, rc(c, k)) for c in range(4 ** k)})
            rank = {r: i for i, r in enumerate(reps)}
            mp = [base + rank[min(c, rc(c, k))] for c in range(4 ** k)]
            base += len(reps)
        else:
            mp = [base + c for c in range(4 ** k)]
            base += 4 ** k
        maps.append(mp)
    rowptr, col, val = [0], [], []
    for s in seqs:
        s = s.lower(); cnt = {}
        for ki, k in enumerate(range(M, N + 1)):
            for p in range(len(s) - k + 1):
                w = s[p:p + k]
                if all(ch in code for ch in w):
                    v = 0
                    for ch in w:
                        v = v * 4 + code[ch]
                    j = maps[ki][v]
                    cnt[j] = cnt.get(j, 0) + 1
        keys = sorted(cnt)
        col += [0] + [j + 1 for j in keys]
        val += [1.0] + [1.0 if binarize else float(cnt[j]) for j in keys]
        rowptr.append(len(col))
    return O.Csr(len(seqs), base + 1, rowptr, col, val)


@pytest.mark.parametrize("M,N,revcomp,binarize", [(4, 4, False, False), (2, 6, True, False), (3, 5, False, True), (1, 3, True, True)])
def test_bundled_fasta_counts(ctx, M, N, revcomp, binarize):
    seqs = read_fasta(os.path.join(GOLDEN, "kmerLr_test_fg.fa")) + read_fasta(os.path.join(GOLDEN, "kmerLr_test_bg.fa"))
    labels = np.array([1] * 11 + [0] * 11, dtype=np.uint8)
    x = host_counts(seqs, M, N, revcomp, binarize)
    m = K.Matrix.from_sequences(ctx, seqs, labels, M, N, revcomp, binarize)
    assert (m.n, m.dim, m.nnz) == (x.n, x.dim, x.nnz)
    rp, col, val = m.to_csr()
    np.testing.assert_array_equal(rp, x.rowptr); np.testing.assert_array_equal(col, x.col); np.testing.assert_array_equal(val, x.val)
    # the CSC side: column sub-select must equal the oracle's
    cols = np.unique(np.r_[0, x.col[x.col % 7 == 3][:30]])
    red = m.select_columns(cols); ro = O.select_columns(x, cols)
    r2, c2, v2 = red.to_csr()
    np.testing.assert_array_equal(r2, ro.rowptr); np.testing.assert_array_equal(c2, ro.col); np.testing.assert_array_equal(v2, ro.val)
    theta = np.zeros(x.dim); theta[0] = 0.1; theta[cols[1:4]] = [0.2, -0.1, 0.05]
    g = K.gradient_dense(ctx, m, theta); go = O.gradient(x, labels, theta)
    assert np.max(np.abs(g - go)) <= 1e-12 * np.max(np.abs(go))


def test_ambiguous_characters_and_short_sequences(ctx):
    seqs = ["ACGTNACGT", "acgtacgtacgt", "AC", "NNNN", "ACGRTTTT", ""]
    labels = np.array([1, 0, 1, 0, 1, 0], dtype=np.uint8)
    for rc in (False, True):
        x = host_counts(seqs, 2, 4, rc)
        m = K.Matrix.from_sequences(ctx, seqs, labels, 2, 4, rc)
        rp, col, val = m.to_csr()
        np.testing.assert_array_equal(rp, x.rowptr); np.testing.assert_array_equal(col, x.col); np.testing.assert_array_equal(val, x.val)


@pytest.fixture()
def host_ingest(monkeypatch):
    # klr_mat_from_csr builds large count matrices on the device too; KLR_HOST_INGEST=1 selects the host builder
    monkeypatch.setenv("KLR_HOST_INGEST", "1")


def test_from_csr_device_route_equals_host_route(ctx, monkeypatch):
    from oracle import oracle as O
    x, labels = O.synth_csr(77, 0, 9000, 9000, 5000, 200, 8)
    v = x.val.copy(); v[5::31] = 300.0; v[x.rowptr[:-1]] = 1.0          # some counts above 255
    md = K.Matrix.from_csr(ctx, x.n, x.dim, x.rowptr, x.col, v, labels)
    monkeypatch.setenv("KLR_HOST_INGEST", "1")
    mh = K.Matrix.from_csr(ctx, x.n, x.dim, x.rowptr, x.col, v, labels)
    monkeypatch.delenv("KLR_HOST_INGEST")
    # the device route builds the row-streamed q64 stream, the host route the tiled formats: different packings of the
    # same integers -- the sweep must agree bit for bit
    assert md.layout()["tiled"] and mh.layout()["tiled"] and md.layout()["format"].startswith("q64")
    for a, b in zip(md.to_csr(), mh.to_csr()):
        np.testing.assert_array_equal(a, b)
    cw = md.class_weights()
    rd = K.sweep_select(ctx, md, None, None, np.zeros(1), cw, 20, want_g=True)
    rh = K.sweep_select(ctx, mh, None, None, np.zeros(1), cw, 20, want_g=True)
    np.testing.assert_array_equal(rd["g"], rh["g"])
    cols = np.nonzero(rd["b"])[0]
    for a, b in zip(md.select_columns(cols).to_csr(), mh.select_columns(cols).to_csr()):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(K.transform_fit(ctx, md, "variance-scaler")[0], K.transform_fit(ctx, mh, "variance-scaler")[0])


def test_large_device_ingest_equals_host_ingest(ctx, host_ingest):
    """Tiled layout built on the device: bit-identical gradient, selection and solve to the host-ingested matrix."""
    rng = np.random.default_rng(5)
    n, L = 6000, 400
    motif_fg, motif_bg = "acgtacg", "ttgacca"
    seqs = []
    for i in range(n):
        s = rng.integers(0, 4, size=L)
        t = "".join("acgt"[v] for v in s)
        if rng.uniform() < 0.6:
            p = int(rng.integers(0, L - 7)); mo = motif_fg if i < n // 2 else motif_bg
            t = t[:p] + mo + t[p + 7:]
        if i % 97 == 0:
            t = "a" * L                                   # homopolymer: counts > 255 (split entries)
        seqs.append(t)
    labels = (np.arange(n) < n // 2).astype(np.uint8)
    x = host_counts(seqs[:40], 6, 6)                     # spot-check the counts against Python on a few rows
    md = K.Matrix.from_sequences(ctx, seqs, labels, 6, 6)
    rp, col, val = md.to_csr()
    np.testing.assert_array_equal(rp[:41], x.rowptr); np.testing.assert_array_equal(col[:x.nnz], x.col)
    np.testing.assert_array_equal(val[:x.nnz], x.val)
    assert val.max() > 255
    mh = K.Matrix.from_csr(ctx, md.n, md.dim, rp, col, val, labels)
    assert md.layout()["tiled"] and mh.layout()["tiled"]
    cw = md.class_weights()
    rd = K.sweep_select(ctx, md, None, None, np.zeros(1), cw, 10, want_g=True)
    rh = K.sweep_select(ctx, mh, None, None, np.zeros(1), cw, 10, want_g=True)
    np.testing.assert_array_equal(rd["g"], rh["g"])                    # exact integer sweep: bitwise equal
    np.testing.assert_array_equal(rd["b"], rh["b"]); assert rd["lam"] == rh["lam"]
    cols = np.nonzero(rd["b"])[0]
    a, _ = K.cd_solve(ctx, md.select_columns(cols), np.zeros(len(cols)), cw=cw, l1reg=rd["lam"] * n)
    b, _ = K.cd_solve(ctx, mh.select_columns(cols), np.zeros(len(cols)), cw=cw, l1reg=rd["lam"] * n)
    np.testing.assert_array_equal(a, b)
    # the planted motifs' 6-mers rank at the top
    top = np.argsort(-np.abs(rd["g"][1:]))[:4]
    names = {"".join("acgt"[(c >> (2 * (5 - i))) & 3] for i in range(6)) for c in top}
    assert names & {"acgtac", "cgtacg", "ttgacc", "tgacca"}
