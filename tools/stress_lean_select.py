"""Dev tool (GPU): hinted selection with the lean conversion (candidates only, sweep_fix_emit on a failed hint) against
KLR_SELECT_NO_LEAN over a random walk of models -- every lambda, count and mask must agree.
python tools/stress_lean_select.py [iterations]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmerlr_amd as K  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(5)
n, m = 30000, 16384
rowptr, col, val, labels = K.synth_csr(77, 0, n, n, m, 300, 40)
ctxs = [K.Context(0), K.Context(0)]
ctxs[1].set_option("KLR_SELECT_NO_LEAN", "1")
mats = [K.Matrix.from_csr(c, n, m + 1, rowptr, col, val, labels) for c in ctxs]
cw = mats[0].class_weights()
fails = 0
cols = None
reds = [None, None]
theta = np.zeros(1)
for it in range(iters):
    N = int(rng.choice([20, 20, 20, 50, 7]))
    if it % 9 == 0 or cols is None:
        r0 = [K.sweep_select(c, mt, None, None, np.zeros(1), cw, N) for c, mt in zip(ctxs, mats)]
        cols = np.nonzero(r0[0]["b"])[0]
        for k in range(2):
            if reds[k] is not None:
                reds[k].close()
            reds[k] = mats[k].select_columns(cols)
        theta = np.zeros(len(cols))
    # a random model on the selected columns: small steps most of the time (hint holds), big jumps sometimes (hint fails)
    scale = 0.02 if rng.random() < 0.7 else float(rng.choice([1.0, 5.0, 30.0]))
    theta = theta * (rng.random() < 0.8) + scale * rng.standard_normal(len(cols))
    out = [K.sweep_select(c, mt, rd, cols, theta, cw, N) for c, mt, rd in zip(ctxs, mats, reds)]
    a, b = out
    same = a["lam"] == b["lam"] and int(a["c"]) == int(b["c"]) and np.array_equal(a["b"], b["b"]) and a["ok"] == b["ok"] and (a["loss"] == b["loss"] or (a["loss"] != a["loss"] and b["loss"] != b["loss"]))
    if not same:
        fails += 1
        print("iteration %d: N %d lam %r / %r c %d / %d" % (it, N, a["lam"], b["lam"], a["c"], b["c"]), flush=True)
print("%d iterations, %d disagreements" % (iters, fails))
sys.exit(1 if fails else 0)
