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
def host_ingest(setopt):
    # klr_mat_from_csr builds large count matrices on the device too; KLR_HOST_INGEST=1 selects the host builder
    setopt("KLR_HOST_INGEST", "1")


def test_from_csr_device_route_equals_host_route(ctx, setopt):
    from oracle import oracle as O
    x, labels = O.synth_csr(77, 0, 9000, 9000, 5000, 200, 8)
    v = x.val.copy(); v[5::31] = 300.0; v[x.rowptr[:-1]] = 1.0          # some counts above 255
    md = K.Matrix.from_csr(ctx, x.n, x.dim, x.rowptr, x.col, v, labels)
    setopt("KLR_HOST_INGEST", "1")
    mh = K.Matrix.from_csr(ctx, x.n, x.dim, x.rowptr, x.col, v, labels)
    setopt("KLR_HOST_INGEST", None)
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


def test_kmers1_known_answers_gapped_revcomp(ctx):
    """TestKmers1 (kmerLr_test.go:30-66), the only counting known-answers the reference holds: M = 4, N = 8, revcomp,
    gapped nucleotide alphabet on kmerLr_test.fa; row 0: gntanc|gntanc = 3, gntcaa|ttganc = 0, aaagaaa|tttcttt = 1,
    aagannt|anntctt = 7, aacgcgna|tncgcgtt = 1, tgaatgca|tgcattca = 1; the pair feature (gntanc, aagannt) = 21.
    Classes are addressed by NAME here; that the reference's six INDICES (4671, 5486, ... among the observed classes) come out
    of this library's class order too is tests/test_kmers_cpu.py."""
    seqs, cur = [], []
    for line in open(os.path.join(GOLDEN, "kmerLr_test.fa")):
        line = line.strip()
        if line.startswith(">"):
            if cur:
                seqs.append("".join(cur)); cur = []
        elif line:
            cur.append(line)
    seqs.append("".join(cur))
    assert len(seqs) == 2
    kc = K.KmerClasses(4, 8, revcomp=True, alphabet="gapped-nucleotide")
    known = {"gntanc|gntanc": 3, "gntcaa|ttganc": 0, "aaagaaa|tttcttt": 1, "aagannt|anntctt": 7, "aacgcgna|tncgcgtt": 1,
             "tgaatgca|tgcattca": 1}
    cls = {}
    for name in known:
        c = kc.find(name)
        assert c >= 0 and kc.name(c) == name and kc.find(name.split("|")[1]) == c
        cls[name] = c
    labels = np.array([1, 0], dtype=np.uint8)
    m = K.Matrix.from_sequences(ctx, seqs, labels, classes=kc)
    assert m.dim == len(kc) + 1
    rp, col, val = m.to_csr()
    row0 = dict(zip(col[rp[0]:rp[1]], val[rp[0]:rp[1]]))
    for name, cnt in known.items():
        assert row0.get(cls[name] + 1, 0.0) == cnt, name
    assert row0[cls["gntanc|gntanc"] + 1] * row0[cls["aagannt|anntctt"] + 1] == 21
    # every row against a brute-force matcher for a few hundred random classes of every length
    import re
    comp = {"a": "t", "c": "g", "g": "c", "t": "a", "n": "n"}
    rng = np.random.default_rng(3)
    for c in rng.choice(len(kc), size=300, replace=False):
        a, b = kc.name(int(c)).split("|")
        assert b == "".join(comp[ch] for ch in reversed(a))
        for i, sq in enumerate(seqs):
            sq = sq.lower()
            cnt = len(re.findall("(?=%s)" % a.replace("n", "[acgt]"), sq))
            if a != b:
                cnt += len(re.findall("(?=%s)" % b.replace("n", "[acgt]"), sq))
            row = dict(zip(col[rp[i]:rp[i + 1]], val[rp[i]:rp[i + 1]]))
            assert row.get(int(c) + 1, 0.0) == cnt, (a, b, i)
    # max-ambiguous = 0 leaves the ungapped classes: the plain --revcomp counter
    kc0 = K.KmerClasses(4, 6, revcomp=True, alphabet="gapped-nucleotide", max_ambiguous=[0])
    kcp = K.KmerClasses(4, 6, revcomp=True)
    assert len(kc0) == len(kcp)
    m0 = K.Matrix.from_sequences(ctx, seqs, labels, classes=kc0)
    mp = K.Matrix.from_sequences(ctx, seqs, labels, 4, 6, revcomp=True)
    for a, b in zip(m0.to_csr(), mp.to_csr()):
        np.testing.assert_array_equal(a, b)


def test_complement_and_reverse_classes(ctx):
    """--complement / --reverse (kmerLr_learn.go:118,121): classes are orbits under the enabled operations; counts
    against a brute-force matcher."""
    rng = np.random.default_rng(8)
    seqs = ["".join("acgt"[v] for v in rng.integers(0, 4, size=300)) for _ in range(6)]
    labels = np.array([1, 1, 1, 0, 0, 0], dtype=np.uint8)
    comp = {"a": "t", "c": "g", "g": "c", "t": "a"}
    for kw in ({"complement": True}, {"reverse": True}, {"complement": True, "reverse": True}):
        kc = K.KmerClasses(3, 4, **kw)
        m = K.Matrix.from_sequences(ctx, seqs, labels, classes=kc)
        rp, col, val = m.to_csr()
        for c in range(len(kc)):
            members = set(kc.name(c).split("|"))
            a = kc.name(c).split("|")[0]
            want = {a}
            if kw.get("complement"):
                want.add("".join(comp[ch] for ch in a))
            if kw.get("reverse"):
                want.add(a[::-1])
            if len(kw) == 2:
                want.add("".join(comp[ch] for ch in a[::-1]))
            assert members == want
            for i, sq in enumerate(seqs):
                cnt = sum(sq[p:p + len(a)] in members for p in range(len(sq) - len(a) + 1))
                row = dict(zip(col[rp[i]:rp[i + 1]], val[rp[i]:rp[i + 1]]))
                assert row.get(c + 1, 0.0) == cnt


def test_kmers2_counter_restricted_to_a_class_list(ctx):
    """TestKmers2 (kmerLr_test.go:70-97): a counter rebuilt from the class list a first pass observed
    (NewKmerCounter(..., data1.Kmers...)) gives every sequence the same counts.  Here the list travels by NAME
    (KmerClass.String() -> klr_kmer_class_find, what ImportConfig does) and the restriction is the column sub-select."""
    seqs, cur = [], []
    for line in open(os.path.join(GOLDEN, "kmerLr_test.fa")):
        line = line.strip()
        if line.startswith(">"):
            if cur:
                seqs.append("".join(cur)); cur = []
        elif line:
            cur.append(line)
    seqs.append("".join(cur))
    labels = np.array([1, 0], dtype=np.uint8)
    kc = K.KmerClasses(4, 8, revcomp=True, alphabet="gapped-nucleotide")
    m = K.Matrix.from_sequences(ctx, seqs, labels, classes=kc)
    rp, col, val = m.to_csr()
    observed = np.unique(col[col > 0]) - 1                     # data1.Kmers: the classes seen in either sequence
    observed = np.sort(np.random.default_rng(5).choice(observed, size=min(3000, len(observed)), replace=False))   # a sample keeps the name round trips short
    names = [kc.name(int(c)) for c in observed]
    kc2 = K.KmerClasses(4, 8, revcomp=True, alphabet="gapped-nucleotide")
    again = np.array([kc2.find(nm) for nm in names])            # the list as a second counter resolves it
    np.testing.assert_array_equal(again, observed)
    red = m.select_columns(np.concatenate(([0], again + 1)))
    rr, rc, rv = red.to_csr()
    assert red.dim == len(observed) + 1
    for i in range(2):
        full = np.zeros(m.dim); full[col[rp[i]:rp[i + 1]]] = val[rp[i]:rp[i + 1]]
        sub = np.zeros(red.dim); sub[rc[rr[i]:rr[i + 1]]] = rv[rr[i]:rr[i + 1]]
        np.testing.assert_array_equal(sub[1:], full[observed + 1])
        assert sub[0] == full[0] == 1.0


def test_iupac_alphabet_counts_by_brute_force(ctx):
    """--alphabet=iupac-nucleotide: every code matches its nucleotide set, ambiguous codes (anything but a c g t) only at
    inner positions and at most max-ambiguous of them; complements r<->y, k<->m, b<->v, d<->h.  The reference holds no
    fixture for this alphabet (parity unpinned): counts against a regular-expression matcher."""
    import re
    sets = {"a": "a", "c": "c", "g": "g", "t": "t", "r": "ag", "y": "ct", "s": "cg", "w": "at", "k": "gt", "m": "ac",
            "b": "cgt", "d": "agt", "h": "act", "v": "acg", "n": "acgt"}
    comp = {"a": "t", "c": "g", "g": "c", "t": "a", "r": "y", "y": "r", "s": "s", "w": "w", "k": "m", "m": "k",
            "b": "v", "v": "b", "d": "h", "h": "d", "n": "n"}
    rng = np.random.default_rng(12)
    seqs = ["".join("acgt"[v] for v in rng.integers(0, 4, size=90)) for _ in range(4)]
    labels = np.array([1, 1, 0, 0], dtype=np.uint8)
    kc = K.KmerClasses(3, 5, revcomp=True, alphabet="iupac-nucleotide", max_ambiguous=[2])
    m = K.Matrix.from_sequences(ctx, seqs, labels, classes=kc)
    assert m.dim == len(kc) + 1
    rp, col, val = m.to_csr()
    rows = [np.zeros(m.dim) for _ in seqs]
    for i in range(len(seqs)):
        rows[i][col[rp[i]:rp[i + 1]]] = val[rp[i]:rp[i + 1]]
    for c in rng.choice(len(kc), size=400, replace=False):
        a, b = kc.name(int(c)).split("|")
        assert b == "".join(comp[ch] for ch in reversed(a)) and a[0] in "acgt" and a[-1] in "acgt"
        assert sum(ch not in "acgt" for ch in a) <= 2 and kc.find(b) == c
        pa = "".join("[%s]" % sets[ch] for ch in a); pb = "".join("[%s]" % sets[ch] for ch in b)
        for i, sq in enumerate(seqs):
            # a window counts once per member it matches (the gapped counter's rule, test_kmers1_known_answers_gapped_revcomp)
            cnt = len(re.findall("(?=%s)" % pa, sq)) + (len(re.findall("(?=%s)" % pb, sq)) if a != b else 0)
            assert rows[i][int(c) + 1] == cnt, (a, b, i)
    # no ambiguous letters allowed: the plain --revcomp counter
    m0 = K.Matrix.from_sequences(ctx, seqs, labels, classes=K.KmerClasses(3, 5, revcomp=True, alphabet="iupac-nucleotide", max_ambiguous=[0]))
    mp = K.Matrix.from_sequences(ctx, seqs, labels, 3, 5, revcomp=True)
    for x, y in zip(m0.to_csr(), mp.to_csr()):
        np.testing.assert_array_equal(x, y)
