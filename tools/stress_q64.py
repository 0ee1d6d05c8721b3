"""Dev tool (GPU): random count matrices (shapes, clustered columns, heavy counts, near-empty rows) through the q64 formats and the
tiled u32 format: the gradients must agree bit for bit.  SEED=.. ITERS=.. python tools/stress_q64.py"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmerlr_amd as K
ctx = K.Context(0)
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
bad = 0
for it in range(int(os.environ.get("ITERS", "24"))):
    n = int(rng.integers(2048, 30000))
    dim = int(rng.choice([3000, 4033, 8129, 20000, 70000, 300000, 1200000]))
    per_row = int(rng.integers(max(40, (1 << 20) // n + 8), max(60, (1 << 20) // n + 200)))
    per_row = min(per_row, dim // 2)
    if n * per_row < (1 << 20) + 4 * n:
        continue
    cols = np.sort(rng.integers(1, dim, size=(n, per_row)), axis=1)
    if rng.uniform() < 0.3:                                  # clustered columns: long runs in few strips
        cols = np.sort((cols % max(dim // 8, 64)) + 1, axis=1)
    keep = np.ones_like(cols, dtype=bool); keep[:, 1:] = cols[:, 1:] != cols[:, :-1]
    heavy = rng.uniform() < 0.4
    counts = (rng.integers(1, 4, size=(n, per_row)) if not heavy else rng.choice([1, 1, 1, 2, 7, 40, 300], size=(n, per_row))).astype(np.float64)
    if rng.uniform() < 0.3:                                  # some rows nearly empty
        kill = rng.uniform(size=n) < 0.5
        keep[kill, 2:] = False
    rowlen = 1 + keep.sum(axis=1)
    rowptr = np.zeros(n + 1, dtype=np.int64); np.cumsum(rowlen, out=rowptr[1:])
    if rowptr[-1] < (1 << 20) + 16:
        continue
    col = np.zeros(rowptr[-1], dtype=np.int64); val = np.ones(rowptr[-1])
    pos = (rowptr[:-1, None] + np.cumsum(keep, axis=1))[keep]
    col[pos] = cols[keep]; val[pos] = counts[keep]
    labels = (rng.uniform(size=n) < 0.5).astype(np.uint8)
    theta = np.zeros(dim); theta[0] = 0.2
    idx = 1 + rng.choice(dim - 1, size=20, replace=False); theta[idx] = rng.normal(size=20) * 0.05
    gs = {}
    for fmt in ("u32", "q64c12", "q64c13"):
        os.environ["KLR_ENTRY_FORMAT"] = fmt; ctx.reload_env()
        m = K.Matrix.from_csr(ctx, n, dim, rowptr, col, val, labels)
        lay = m.layout()
        cw = m.class_weights()
        gs[fmt] = (K.gradient_dense(ctx, m, theta, cw=cw), lay["format"], lay["nseg"])
        m.close()
    ok = all(np.array_equal(gs["u32"][0], gs[f][0]) for f in ("q64c12", "q64c13"))
    print("it %2d n %6d dim %8d per_row %4d heavy %d formats %s pieces %s %s" % (it, n, dim, per_row, heavy, [gs[f][1] for f in gs], [gs[f][2] for f in gs], "ok" if ok else "MISMATCH"), flush=True)
    bad += not ok
print("mismatches", bad)
