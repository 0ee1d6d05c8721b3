"""GPU: the tie contract of the selection on count data (README "Parity contract", DESIGN.md "Selection at ties").

At theta = 0 the gradient of a k-mer count matrix is an integer column-sum difference over 2n, so many |g_j| are
mathematically EQUAL.  The reference's sequential f64 loop leaves such ties an ulp apart in an order-dependent way
and its unstable sort.Sort picks among them arbitrarily; its lambda = (v + w)/2 then takes w = v minus an ulp.
The device gradient is exact integer arithmetic, so equal sums are equal numbers and the contract is:

  1. columns are taken by |g| descending, then column index ascending;
  2. lambda is the midpoint between the N-th largest |g| and the next DISTINCT value below it;
  3. both are functions of the exact column sums only: the same on 1 or many ranks, run to run, for any entry format.

This test pins 1-3 against exact integer sums computed with numpy, and records how far the reference-order
(oracle) result is from it: same columns outside the tie group at the threshold, lambda within half the gap
between the two adjacent distinct values."""
import numpy as np
import pytest

import kmerlr_amd as K
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = K.Context(0)
    yield c
    c.close()


def exact_d(x, labels):
    """bg column sum - fg column sum (exact int64): g_j(theta = 0, cw = (1, 1)) = d_j / (2n)."""
    rows = np.repeat(np.arange(x.n), np.diff(x.rowptr))
    sign = np.where(labels[rows] != 0, -1.0, 1.0)
    d = np.bincount(x.col, weights=sign * x.val, minlength=x.dim)
    assert np.all(d == np.rint(d))
    return d.astype(np.int64)


def rule(d, N):
    a = np.abs(d[1:])
    order = np.lexsort((np.arange(len(a)), -a))
    v = int(a[order[N - 1]])
    w = int(a[a < v].max())
    return np.sort(order[:N]), v, w


@pytest.mark.parametrize("n,m,draws,N,S", [(3000, 8192, 400, 150, 300), (5000, 4096, 300, 200, 80), (6000, 16384, 200, 300, 100)])
def test_tie_contract_on_count_data(ctx, n, m, draws, N, S):
    x, labels = O.synth_csr(17, 0, n, n, m, draws, S)
    d = exact_d(x, labels)
    a = np.abs(d[1:])
    picked, v, w = rule(d, N)
    # the data really ties across the threshold: more columns share the N-th value than fit
    tied = np.nonzero(a == v)[0]
    assert len(tied) >= 2 and np.sum(a >= v) > N, "workload does not tie at the threshold; pick another seed"
    mat = K.Matrix.from_csr(ctx, x.n, x.dim, x.rowptr, x.col, x.val, labels)
    assert mat.layout()["tiled"]
    r = K.sweep_select(ctx, mat, None, None, np.zeros(1), (1.0, 1.0), N, want_g=True)
    # (1) the picked set: |g| descending, column ascending, evaluated on the exact sums
    np.testing.assert_array_equal(np.nonzero(r["b"])[0][1:] - 1, picked)
    # equal sums -> bitwise equal gradients
    assert np.all(np.abs(r["g"][1:][tied]) == np.abs(r["g"][1:][tied[0]]))
    # (2) lambda: the exact midpoint between DISTINCT values
    lam_exact = (v / (2.0 * n) + w / (2.0 * n)) / 2.0
    assert abs(r["lam"] - lam_exact) <= 1e-13 * lam_exact
    # (3) the oracle's rule applied to the device gradient reproduces the device result bit for bit
    b0 = np.zeros(x.dim, dtype=np.uint8); b0[0] = 1
    ok, lam_o, c_o, b_o = O.select(r["g"][1:], b0, 0, N)
    assert lam_o == r["lam"] and c_o == r["c"]
    np.testing.assert_array_equal(b_o, r["b"])
    # ---- the reference-order result, recorded beside it: f64 accumulation in row order -----------------------
    g_ref = O.gradient(x, labels, np.zeros(x.dim))
    assert np.max(np.abs(g_ref - r["g"])) <= 1e-14                           # the gradients agree to an ulp of max|g| ...
    ok_r, lam_r, c_r, b_r = O.select(g_ref[1:], b0, 0, N)
    # ... but the rule is discontinuous at a tie: the reference's w is an ulp-perturbed member of the tie group
    assert lam_exact - 1e-15 <= lam_r <= v / (2.0 * n) + 1e-15               # between the true midpoint and v
    assert abs(lam_r - r["lam"]) <= (v - w) / (4.0 * n) + 1e-15              # at most half the gap of adjacent distinct values
    diff = np.setxor1d(np.nonzero(b_r)[0], np.nonzero(r["b"])[0]) - 1
    assert np.all(a[diff] == v)                                              # the sets differ only inside the tie group at v
    print("tie contract: n=%d N=%d v=%d w=%d tied=%d lambda device %.17g (exact midpoint) reference-order %.17g differing columns %d"
          % (n, N, v, w, len(tied), r["lam"], lam_r, len(diff)))
    mat.close()


def test_tie_contract_whole_path(ctx):
    """The leapfrog path on tied count data (no de-tying scale): the first lambda is the exact midpoint, the run is
    bitwise reproducible, the classifier has exactly N nonzero coefficients and satisfies the KKT conditions."""
    n, m, N = 3000, 8192, 150
    x, labels = O.synth_csr(17, 0, n, n, m, 400, 2 * N)
    d = exact_d(x, labels)
    picked, v, w = rule(d, N)
    mat = K.Matrix.from_csr(ctx, x.n, x.dim, x.rowptr, x.col, x.val, labels)

    def run():
        est = K.hip_estimator(ctx, max_iterations=4000)
        cls = est.estimate(mat._h, [N])
        ep = est.epochs()
        est.close()
        return cls[0], ep
    cl, ep = run()
    lam0 = (v / (2.0 * n) + w / (2.0 * n)) / 2.0
    assert abs(ep[0][0] - lam0) <= 1e-13 * lam0
    assert int(np.count_nonzero(cl["theta"][1:])) == N
    red = mat.select_columns(cl["cols"])
    g = K.gradient_dense(ctx, red, cl["theta"], (1.0, 1.0))
    nz = cl["theta"][1:] != 0
    assert abs(g[0]) < 1e-4 and np.max(np.abs(np.abs(g[1:][nz]) - cl["lam"])) < 2e-3 * cl["lam"]
    cl2, ep2 = run()
    np.testing.assert_array_equal(cl2["cols"], cl["cols"]); np.testing.assert_array_equal(cl2["theta"], cl["theta"])
    assert ep2 == ep
    red.close(); mat.close()
