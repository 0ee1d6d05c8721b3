"""Dev tool (GPU, KLR_ABLATIONS build): q64 sweep kernel, pipeline depth (A, B) x workgroups per CU, cold and resident."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmerlr_amd as K  # noqa: E402

SHAPES = {"c2": (100000, 65536, 100, "q64c12"), "c3": (125000, 1048576, 500, "q64c13")}
ctx = K.Context(0)
for sh in (sys.argv[1:] or ["c2", "c3"]):
    n, m, N, fmt = SHAPES[sh]
    os.environ["KLR_ENTRY_FORMAT"] = fmt
    rowptr, col, val, labels = K.synth_csr(20261004, 0, n, n, m, 993, 2 * N)
    mats = {}
    for w in (1, 2):
        os.environ["KLR_Q64_WGS"] = str(w)
        mats[w] = K.Matrix.from_csr(ctx, n, m + 1, rowptr, col, val, labels)
    del rowptr, col, val
    cw = mats[1].class_weights()
    combos = [(w, d) for w in (1, 2) for d in (31, 52, 73, 115, 71)]
    res = {c: [] for c in combos}
    for r in range(4):
        for (w, d) in combos:
            os.environ["KLR_Q64_DEPTH"] = str(d)
            ctx.prof_enable(True); ctx.prof_read(reset=True)
            for _ in range(6):
                ctx.evict(1 << 30)
                K.sweep_select(ctx, mats[w], None, None, np.zeros(1), cw, N)
            nl, ms = ctx.prof_read(reset=True)
            ctx.prof_enable(False)
            if r:
                res[(w, d)].append(ms / nl * 1e3)
    for c in combos:
        print("%s wgs/cu %d depth %3d  cold %7.2f us" % (sh, c[0], c[1], np.median(res[c])), flush=True)
    for w in mats:
        mats[w].close()
