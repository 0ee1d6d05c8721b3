"""GPU: `kmerlr --ranks=N learn` (the multi-GPU entry point, kmerLr_learn.go:71-108 over row shards) with two ranks
on ONE card through the host-staged shared-memory transport, against the one-rank run of the same command line; the
`learn` forms the reference has -- warm start from a model, --max-samples, --validation-size, `--opt value` -- and
the refusal of unknown options."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "kmerlr_amd", "_lib", "kmerlr")
FG, BG = os.path.join(GOLDEN, "kmerLr_test_fg.fa"), os.path.join(GOLDEN, "kmerLr_test_bg.fa")
SFG, SBG = os.path.join(GOLDEN, "scoresLr_test_fg.table"), os.path.join(GOLDEN, "scoresLr_test_bg.table")


def run(args, cwd, ok=True):
    r = subprocess.run([CLI] + args, cwd=cwd, capture_output=True, text=True, timeout=600)
    if ok:
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r


def model(path):
    m = json.load(open(path))
    return m["Parameters"], np.array(m["Distributions"][0]["Parameters"])


def same_model(a, b, rtol=1e-9):
    """Everything integer -- the counter, the k-mer / feature lists -- is identical text; the coefficients agree to
    rtol (the IRLS column sums are f64 and their order depends on the sharding)."""
    pa, ta = model(a)
    pb, tb = model(b)
    assert pa == pb
    np.testing.assert_allclose(ta, tb, rtol=rtol, atol=1e-13)


def test_two_ranks_bundled_fasta_equals_one_rank(tmp_path):
    """BASELINE configs[1] on two ranks: same selected 4-mers, same path, theta to 1e-9; only rank 0 writes and speaks."""
    common = ["learn", "--lambda-auto=10", "--save-trace", "--save-path", "4", "4", FG, BG]
    r1 = run(["-v"] + common + ["one"], tmp_path)
    r2 = run(["-v", "--ranks=2", "--transport=shm"] + common + ["two"], tmp_path)
    same_model(tmp_path / "one.json", tmp_path / "two.json")
    l1 = [ln for ln in r1.stderr.splitlines() if ln.startswith("Estimating parameters")]
    l2 = [ln for ln in r2.stderr.splitlines() if ln.startswith("Estimating parameters")]
    assert l1 == l2 and len(l1) >= 2                                  # the lambda sequence, printed once (rank 0)
    assert r2.stderr.count("Exporting model to `two.json'") == 1
    assert "2 ranks over shm" in r2.stderr
    t1 = open(tmp_path / "one.trace").read().splitlines()
    t2 = open(tmp_path / "two.trace").read().splitlines()
    assert len(t1) == len(t2)
    p1 = open(tmp_path / "one.path").read().splitlines()
    p2 = open(tmp_path / "two.path").read().splitlines()
    assert len(p1) == len(p2) and [ln.split()[0] for ln in p1] == [ln.split()[0] for ln in p2]      # lambda column


def test_two_ranks_scores_table_and_cv(tmp_path):
    """scoresLr tables on two ranks (8 rows each: one class per rank), plain learn and 4-fold CV: the joined prediction
    table holds every row once, in fold order, equal to the one-rank table."""
    run(["--type=scoresLr", "learn", "--lambda-auto=2", SFG, SBG, "one"], tmp_path)
    run(["--type=scoresLr", "--ranks=2", "--transport=shm", "learn", "--lambda-auto=2", SFG, SBG, "two"], tmp_path)
    same_model(tmp_path / "one.json", tmp_path / "two.json")
    p, _ = model(tmp_path / "two.json")
    assert p["Index"] == [1, 6]                                        # scoresLr_test.go:53-58
    cv = ["learn", "--lambda-auto=2,3", "--k-fold-cv=4", "--save-trace", SFG, SBG]
    run(["--type=scoresLr"] + cv + ["cv1"], tmp_path)
    run(["--type=scoresLr", "--ranks=2", "--transport=shm"] + cv + ["cv2"], tmp_path)
    for la in (2, 3):
        a = np.loadtxt(tmp_path / ("cv1_%d.table" % la), skiprows=1)
        b = np.loadtxt(tmp_path / ("cv2_%d.table" % la), skiprows=1)
        assert a.shape == (16, 2) and b.shape == (16, 2)
        np.testing.assert_array_equal(a[:, 1], b[:, 1])
        np.testing.assert_allclose(a[:, 0], b[:, 0], rtol=1e-6, atol=1e-9)
        la_, lb_ = (np.loadtxt(tmp_path / ("cv%d_%d_loss.table" % (k, la)), skiprows=1) for k in (1, 2))
        np.testing.assert_allclose(la_, lb_, rtol=1e-6)
    for f in range(1, 5):
        same_model(tmp_path / ("cv1_cv%d_2.json" % f), tmp_path / ("cv2_cv%d_2.json" % f), rtol=1e-7)
        assert os.path.exists(tmp_path / ("cv2_cv%d.trace" % f))      # learn_parameters: one _cv<f> file set per fold


def test_two_ranks_ensemble_and_warm_start(tmp_path):
    """--ensemble-size (members = row masks by global row id: every rank masks its own rows) and the warm start from a model
    file on two ranks against one."""
    ens = ["learn", "--lambda-auto=4", "--ensemble-size=3", "--ensemble-summary=mean", "4", "4", FG, BG]
    run(ens + ["e1"], tmp_path)
    run(["--ranks=2", "--transport=shm"] + ens + ["e2"], tmp_path)
    m1, m2 = json.load(open(tmp_path / "e1.json")), json.load(open(tmp_path / "e2.json"))
    assert m1["Name"] == m2["Name"] == "kmerLr [mean]" and m1["Parameters"] == m2["Parameters"]
    assert len(m1["Distributions"]) == len(m2["Distributions"]) == 4                  # three members + the transform
    for a, b in zip(m1["Distributions"][:3], m2["Distributions"][:3]):
        np.testing.assert_allclose(a["Parameters"], b["Parameters"], rtol=1e-8, atol=1e-13)
    run(["learn", "--lambda-auto=5", "--revcomp", "4", "4", FG, BG, "five"], tmp_path)
    run(["learn", "--lambda-auto=9", "five.json", FG, BG, "w1"], tmp_path)
    run(["--ranks=2", "--transport=shm", "learn", "--lambda-auto=9", "five.json", FG, BG, "w2"], tmp_path)
    same_model(tmp_path / "w1.json", tmp_path / "w2.json", rtol=1e-8)


def write_fasta(path, seqs):
    with open(path, "w") as f:
        for i, s in enumerate(seqs):
            f.write(">s%d\n%s\n" % (i, s))


@pytest.fixture(scope="module")
def big_fasta(tmp_path_factory):
    """Count data large enough for the packed integer stream on EACH of two or three shards (n >= 2048 rows, nnz >= 2^20):
    2 x 3,300 sequences of 700 bp, a few planted 6-mers per class."""
    d = tmp_path_factory.mktemp("bigfa")
    rng = np.random.default_rng(20261005)
    motifs = {1: ["acgtgc", "ttgaca", "ggatcc"], 0: ["cccggg", "atatat", "gattac"]}
    out = {}
    for label, name in ((1, "fg.fa"), (0, "bg.fa")):
        seqs = []
        for _ in range(3300):
            s = rng.integers(0, 4, size=700)
            t = "".join("acgt"[c] for c in s)
            for mt in motifs[label]:
                for _k in range(int(rng.integers(0, 4))):
                    p = int(rng.integers(0, 700 - 6))
                    t = t[:p] + mt + t[p + 6:]
            seqs.append(t)
        write_fasta(d / name, seqs)
        out[label] = str(d / name)
    return out


def test_two_ranks_integer_stream_path(tmp_path, big_fasta):
    """Shards that take the packed integer stream: the full-width sums are exact integers, so the lambda of every epoch's
    selection is printed identically and the selected 6-mers are identical; reduce-scatter form of the collective
    forced (KLR_RS_MIN_DIM) on top of the all-reduce run."""
    common = ["learn", "--lambda-auto=6,12", "--revcomp", "--save-path", "6", "6", big_fasta[1], big_fasta[0]]
    r1 = run(["-v"] + common + ["one"], tmp_path)
    r2 = run(["-v", "--ranks=2", "--transport=shm"] + common + ["two"], tmp_path)
    env = dict(os.environ, KLR_RS_MIN_DIM="1")
    r3 = subprocess.run([CLI, "-v", "--ranks=2", "--transport=shm"] + common + ["three"], cwd=tmp_path, capture_output=True, text=True,
                        timeout=600, env=env)
    assert r3.returncode == 0, r3.stderr[-3000:]
    for la in (6, 12):
        same_model(tmp_path / ("one_%d.json" % la), tmp_path / ("two_%d.json" % la))
        same_model(tmp_path / ("two_%d.json" % la), tmp_path / ("three_%d.json" % la), rtol=0)      # same sums, same rank order
    # three ranks (slices of a third, a three-way merge of the candidates), the sweep in two strip ranges with the collective
    # of the first range behind the second range's launch
    env = dict(os.environ, KLR_RS_MIN_DIM="1", KLR_RS="1", KLR_SWEEP_RANGES="2")
    r4 = subprocess.run([CLI, "-v", "--ranks=3", "--transport=shm"] + common + ["four"], cwd=tmp_path, capture_output=True, text=True,
                        timeout=600, env=env)
    assert r4.returncode == 0, r4.stderr[-3000:]
    for la in (6, 12):
        same_model(tmp_path / ("one_%d.json" % la), tmp_path / ("four_%d.json" % la))
    lam = [[ln for ln in r.stderr.splitlines() if ln.startswith("Estimating parameters")] for r in (r1, r2, r3, r4)]
    assert lam[0][0] == lam[1][0] == lam[2][0] == lam[3][0]            # first epoch (theta = 0): exact integer sums


def test_three_ranks_strip_ranges(tmp_path, big_fasta):
    """8-mers (32,896 revcomp classes: nine strips): three ranks, the reduce-scatter form in two strip ranges -- the collective
    of the first range runs behind the second range's launch -- against the one-rank run."""
    import re
    common = ["learn", "--lambda-auto=8", "--revcomp", "8", "8", big_fasta[1], big_fasta[0]]
    r1 = run(["-v"] + common + ["one"], tmp_path)
    env = dict(os.environ, KLR_RS_MIN_DIM="1", KLR_RS="1", KLR_SWEEP_RANGES="2")
    r3 = subprocess.run([CLI, "-v", "--ranks=3", "--transport=shm"] + common + ["three"], cwd=tmp_path, capture_output=True, text=True,
                        timeout=600, env=env)
    assert r3.returncode == 0, r3.stderr[-3000:]
    same_model(tmp_path / "one.json", tmp_path / "three.json")
    lam = [[ln for ln in r.stderr.splitlines() if ln.startswith("Estimating parameters")] for r in (r1, r3)]
    assert lam[0][0] == lam[1][0]
    rep = re.search(r"Sweep collectives: (\d+) all-reduce, (\d+) reduce-scatter \+ all-gather \((\d+) repeated as all-reduce, (\d+) in strip ranges\)", r3.stderr)
    assert rep, r3.stderr[-2000:]
    ar, rs, fb, rg = (int(v) for v in rep.groups())
    assert rs >= 2 and rg == rs and fb <= 1          # every wide sweep in the ranged form; the first one is verified against the all-reduce


def test_unknown_options_and_getopt_spelling(tmp_path):
    r = run(["--type=scoresLr", "learn", "--lambda-auto=2", "--no-such-option", SFG, SBG, "x"], tmp_path, ok=False)
    assert r.returncode != 0 and "unknown option `--no-such-option'" in r.stderr
    r = run(["--type=scoresLr", "learn", "--revcomp", "--lambda-auto=2", SFG, SBG, "x"], tmp_path, ok=False)
    assert r.returncode != 0 and "unknown option `--revcomp' for --type=scoresLr" in r.stderr
    r = run(["--no-such-global", "learn"], tmp_path, ok=False)
    assert r.returncode != 0 and "unknown option" in r.stderr
    # `--opt value` (getopt) and `--opt=value` are the same command
    run(["--type", "scoresLr", "--seed", "1", "learn", "--lambda-auto", "2", "--epsilon-loss", "1e-8", SFG, SBG, "a"], tmp_path)
    run(["--type=scoresLr", "--seed=1", "learn", "--lambda-auto=2", "--epsilon-loss=1e-8", SFG, SBG, "b"], tmp_path)
    assert open(tmp_path / "a.json").read() == open(tmp_path / "b.json").read()
    # SAGA's knobs are the reference's options: accepted, without effect
    run(["--type=scoresLr", "learn", "--lambda-auto=2", "--scale-step-size=0.5", "--adaptive-step-size", "--threads-saga=4", SFG, SBG, "c"], tmp_path)
    assert open(tmp_path / "c.json").read() == open(tmp_path / "a.json").read()


def test_warm_start_from_model(tmp_path):
    """learn MODEL.json FG BG OUT (kmerLr_learn.go:89-92): continuing a --lambda-auto=5 model to 10 features is the second
    half of a --lambda-auto=5,10 run (same estimator state: columns and coefficients; the counter comes from the model)."""
    run(["learn", "--lambda-auto=5,10", "--revcomp", "4", "4", FG, BG, "both"], tmp_path)
    run(["learn", "--lambda-auto=5", "--revcomp", "4", "4", FG, BG, "five"], tmp_path)
    assert open(tmp_path / "five.json").read() == open(tmp_path / "both_5.json").read()
    run(["learn", "--lambda-auto=10", "five.json", FG, BG, "ten"], tmp_path)
    pa, ta = model(tmp_path / "both_10.json")
    pb, tb = model(tmp_path / "ten.json")
    assert pa == pb and pb["Revcomp"] is True and pb["M"] == 4         # the model's counter, not the command line's
    np.testing.assert_allclose(ta, tb, rtol=1e-12, atol=0)
    # scoresLr: [MODEL.json] FG BG BASENAME
    run(["--type=scoresLr", "learn", "--lambda-auto=2,4", SFG, SBG, "s"], tmp_path)
    run(["--type=scoresLr", "learn", "--lambda-auto=4", "s_2.json", SFG, SBG, "s4"], tmp_path)
    pa, ta = model(tmp_path / "s_4.json")
    pb, tb = model(tmp_path / "s4.json")
    assert pa == pb
    np.testing.assert_allclose(ta, tb, rtol=1e-12, atol=0)


def fasta_records(path):
    recs, cur = [], None
    for line in open(path):
        if line.startswith(">"):
            if cur is not None:
                recs.append(cur)
            cur = [line, ""]
        elif cur is not None:
            cur[1] += line
    if cur is not None:
        recs.append(cur)
    return recs


def test_max_samples(tmp_path):
    """reduce_samples (kmerLr_data.go:288-303): the first m1 fg and m2 bg records, m1 = int(n1 / (n1 + n2) * max)."""
    fg, bg = fasta_records(FG), fasta_records(BG)
    n1, n2 = len(fg), len(bg)
    mx = 13
    m1 = int(n1 / (n1 + n2) * mx); m2 = mx - m1
    for name, recs in (("f.fa", fg[:m1]), ("b.fa", bg[:m2])):
        with open(tmp_path / name, "w") as f:
            for h, s in recs:
                f.write(h + s)
    r = run(["-v", "learn", "--lambda-auto=4", "--max-samples=%d" % mx, "4", "4", FG, BG, "red"], tmp_path)
    assert "Reduced training data from (%d,%d) to (%d,%d) samples" % (n1, n2, m1, m2) in r.stderr
    run(["learn", "--lambda-auto=4", "4", "4", str(tmp_path / "f.fa"), str(tmp_path / "b.fa"), "cut"], tmp_path)
    assert open(tmp_path / "red.json").read() == open(tmp_path / "cut.json").read()
    run(["--ranks=2", "--transport=shm", "learn", "--lambda-auto=4", "--max-samples=%d" % mx, "4", "4", FG, BG, "red2"], tmp_path)
    same_model(tmp_path / "red.json", tmp_path / "red2.json")


def test_validation_size_selects_by_validation_loss(tmp_path):
    """--validation-size (getCvGroups / filterCvGroup, kmerLr_crossvalidation.go:56-120; selection of the best classifier
    kmerLr_estimator_ensemble.go:119-135): the run must equal training every lambda_auto on the training rows alone and
    keeping the model with the lowest `kmerlr loss` on the validation rows."""
    import kmerlr_amd as K
    from kmerlr_amd import _driver
    L = _driver.bind_driver(K.lib())
    fg, bg = fasta_records(FG), fasta_records(BG)
    n1, n_all = len(fg), len(fg) + len(bg)
    val = np.zeros(n_all, dtype=np.uint8)
    grp = np.zeros(n_all, dtype=np.int32)
    L.klr_cv_groups_val(n_all, 1, 0.3, 1, 0, n_all, grp.ctypes.data, val.ctypes.data)
    assert np.all(grp == -1) and 0 < val.sum() < n_all
    allrec = fg + bg
    for tag, keep in (("tr", val == 0), ("va", val == 1)):
        for name, lo, hi in (("fg", 0, n1), ("bg", n1, n_all)):
            with open(tmp_path / ("%s_%s.fa" % (tag, name)), "w") as f:
                for i in range(lo, hi):
                    if keep[i]:
                        f.write(allrec[i][0] + allrec[i][1])
    r = run(["-v", "learn", "--lambda-auto=3,8", "--validation-size=0.3", "4", "4", FG, BG, "v"], tmp_path)
    assert "> Selecting classifier" in r.stderr and os.path.exists(tmp_path / "v.json") and not os.path.exists(tmp_path / "v_3.json")
    run(["learn", "--lambda-auto=3,8", "4", "4", str(tmp_path / "tr_fg.fa"), str(tmp_path / "tr_bg.fa"), "t"], tmp_path)
    losses = [float(run(["loss", "t_%d.json" % la, str(tmp_path / "va_fg.fa"), str(tmp_path / "va_bg.fa")], tmp_path).stdout) for la in (3, 8)]
    best = (3, 8)[int(np.argmin(losses))]
    same_model(tmp_path / "v.json", tmp_path / ("t_%d.json" % best), rtol=1e-8)
    # with folds: one model per fold, one joined table
    run(["--type=scoresLr", "learn", "--lambda-auto=2,3", "--k-fold-cv=2", "--validation-size=0.25", SFG, SBG, "w"], tmp_path)
    assert os.path.exists(tmp_path / "w_cv1.json") and os.path.exists(tmp_path / "w_cv2.json") and os.path.exists(tmp_path / "w.table")
    assert np.loadtxt(tmp_path / "w.table", skiprows=1).shape == (16, 2)


def test_rank_failure_ends_the_run(tmp_path):
    """A rank that cannot start (no such device) must end the whole command with an error, not hang it."""
    r = run(["--ranks=2", "--transport=shm", "--devices=0,63", "learn", "--lambda-auto=2", "4", "4", FG, BG, "x"], tmp_path, ok=False)
    assert r.returncode != 0 and "out of range" in r.stderr


def test_ensembles_inside_cross_validation(tmp_path):
    """--ensemble-size with --k-fold-cv (estimate_ensemble runs inside every fold, kmerLr_estimator_ensemble.go:118): one
    ensemble model per fold, a prediction table of summarised member predictions; one rank and two."""
    # (--max-epochs: a member of a fold of the 16-row table trains on 4 rows, possibly of one class, and the leapfrog loop
    # -- unbounded in the reference, kmerLr_estimator.go:225 -- would then chase lambda -> 0 for ever)
    cv = ["learn", "--lambda-auto=2", "--k-fold-cv=2", "--ensemble-size=2", "--ensemble-summary=mean", "--max-epochs=3", SFG, SBG]
    run(["--type=scoresLr"] + cv + ["e1"], tmp_path)
    run(["--type=scoresLr", "--ranks=2", "--transport=shm"] + cv + ["e2"], tmp_path)
    for f in (1, 2):
        m1, m2 = (json.load(open(tmp_path / ("%s_cv%d.json" % (b, f)))) for b in ("e1", "e2"))
        assert m1["Name"] == "scoresLr [mean]" and len(m1["Distributions"]) == 3 and m1["Parameters"] == m2["Parameters"]
        for a, b in zip(m1["Distributions"][:2], m2["Distributions"][:2]):
            np.testing.assert_allclose(a["Parameters"], b["Parameters"], rtol=1e-7, atol=1e-12)
    a = np.loadtxt(tmp_path / "e1.table", skiprows=1)
    b = np.loadtxt(tmp_path / "e2.table", skiprows=1)
    assert a.shape == (16, 2)
    np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-9)
