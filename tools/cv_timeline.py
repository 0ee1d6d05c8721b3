"""Dev tool (GPU): k-fold cross-validation (klr_estimator_crossvalidate) with the folds in turn and at once
(klr_config.threads_cv).  python tools/cv_timeline.py [rows] [features] [N] [folds]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmerlr_amd as K                         # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
N = int(sys.argv[3]) if len(sys.argv) > 3 else 100
k = int(sys.argv[4]) if len(sys.argv) > 4 else 5
ctx = K.Context(0)
rowptr, col, val, labels = K.synth_csr(20240229, 0, n, n, m, 993, 2 * N)
mat = K.Matrix.from_csr(ctx, n, m + 1, rowptr, col, val, labels, n)
ctx.synchronize()
ref = None
for rep in range(2):
    for threads in (1, 2, 3, 4, 5):
        est = K.hip_estimator(ctx, threads_cv=threads)
        t0 = time.perf_counter()
        r = est.crossvalidate(mat._h, n, k, [N], seed=3)
        dt = time.perf_counter() - t0
        if ref is None:
            ref = r
        same = np.array_equal(ref["predictions"], r["predictions"]) and np.array_equal(ref["loss_test"], r["loss_test"])
        print("rep %d: %d-fold CV, %d folds at once: %.1f ms (%.1f ms per fold); results %s" % (rep, k, threads, dt * 1e3, dt * 1e3 / k, "equal" if same else "DIFFER"), flush=True)
        est.close()
