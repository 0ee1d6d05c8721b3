"""Dev tool (GPU): reduced solve on a DENSE n x d panel (scoresLr-style table), Gram on the matrix cores against the sparse
LDS-atomic build.  usage: python tools/dense_gram_timeline.py [n] [d]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmerlr_amd as K  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 101
rng = np.random.default_rng(1)
dense = np.c_[np.ones(n), rng.normal(size=(n, d - 1))]
beta = np.zeros(d); beta[1:9] = rng.normal(size=8)
labels = (rng.uniform(size=n) < 1.0 / (1.0 + np.exp(-dense @ beta))).astype(np.uint8)
rowptr = np.arange(n + 1, dtype=np.int64) * d
col = np.tile(np.arange(d, dtype=np.int64), n)
ctx = K.Context(0)
mat = K.Matrix.from_csr(ctx, n, d, rowptr, col, dense.reshape(-1), labels)
red = mat.select_columns(np.arange(d))
cw = mat.class_weights()
for env in (None, "1"):
    if env:
        os.environ["KLR_GRAM_NO_MFMA"] = env; ctx.reload_env()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        th, recs = K.cd_solve(ctx, red, np.zeros(d), cw=cw, l1reg=0.01 * n, eps_loss=1e-9, max_iter=60)
        best = min(best, time.perf_counter() - t0)
    print("n=%d d=%d %-28s solve %.2f ms, %d sweeps" % (n, d, "sparse Gram (LDS atomics)" if env else "dense-panel MFMA Gram", best * 1e3, len(recs)), flush=True)
