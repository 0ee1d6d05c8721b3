#!/usr/bin/env python3
"""End-to-end leapfrog run (KmerLrEstimator.Estimate, --lambda-auto=N) on the synthetic configs[2] shard, timed.

  python3 tools/estimate_timeline.py [rows] [features] [lambda_auto]

Prints wall seconds of estimate(), epochs (selection steps), CD iterations, and the classifier size.  Run it under
`rocprofv3 --kernel-trace --stats` to see which kernels the wall time goes to.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kmerlr_amd as K                         # noqa: E402
from kmerlr_amd import _driver                 # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    draws = 993
    ctx = K.Context(0)
    rowptr, col, val, labels = K.synth_csr(int(os.environ.get("EST_SEED", "20240229")), 0, n, n, m, draws, 2 * N)
    full = K.Matrix.from_csr(ctx, n, m + 1, rowptr, col, val, labels, n)
    ctx.synchronize()
    out = []
    for rep in range(3):
        est = _driver.hip_estimator(ctx, save_trace=1)
        t0 = time.perf_counter()
        est.estimate(full._h, [N])
        dt = time.perf_counter() - t0
        lam, nz = est.epochs()
        cl = est.classifiers()
        out.append(dict(seconds=round(dt, 5), epochs=len(lam), cd_iterations=len(est.trace()),
                        nonzero=int(np.count_nonzero(cl[0]["theta"][1:])), lam=cl[0]["lam"]))
        est.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
