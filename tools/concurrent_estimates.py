"""Dev tool (GPU): T whole leapfrog runs at once, a context and a matrix each, from T host threads -- what concurrent
folds of a cross-validation would gain on one card.  python tools/concurrent_estimates.py [rows] [features] [N] [Tmax]"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmerlr_amd as K                         # noqa: E402
from kmerlr_amd import _driver                 # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
N = int(sys.argv[3]) if len(sys.argv) > 3 else 100
Tmax = int(sys.argv[4]) if len(sys.argv) > 4 else 5
rowptr, col, val, labels = K.synth_csr(20240229, 0, n, n, m, 993, 2 * N)
ctxs, mats = [], []
for t in range(Tmax):
    c = K.Context(0)
    ctxs.append(c)
    mats.append(K.Matrix.from_csr(c, n, m + 1, rowptr, col, val, labels, n))
    c.synchronize()


def one(t, out):
    est = _driver.hip_estimator(ctxs[t])
    t0 = time.perf_counter()
    est.estimate(mats[t]._h, [N])
    out[t] = time.perf_counter() - t0
    est.close()


for rep in range(2):
    for T in range(1, Tmax + 1):
        out = [0.0] * T
        th = [threading.Thread(target=one, args=(t, out)) for t in range(T)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        wall = time.perf_counter() - t0
        print("rep %d: %d concurrent estimates: wall %.1f ms (%.1f ms per estimate; slowest thread %.1f ms)" % (rep, T, wall * 1e3, wall * 1e3 / T, max(out) * 1e3), flush=True)
