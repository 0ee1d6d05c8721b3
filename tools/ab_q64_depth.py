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
    os.environ["KLR_ENTRY_FORMAT"] = fmt; ctx.reload_env()
    rowptr, col, val, labels = K.synth_csr(20261004, 0, n, n, m, 993, 2 * N)
    mats = {}
    for w in (int(v) for v in os.environ.get("AB_WGS", "1").split(",")):
        os.environ["KLR_Q64_WGS"] = str(w); ctx.reload_env()
        mats[w] = K.Matrix.from_csr(ctx, n, m + 1, rowptr, col, val, labels)
    del rowptr, col, val
    cw = mats[1].class_weights()
    combos = [(w, d) for w in mats for d in (int(v) for v in os.environ.get("AB_DEPTHS", "10,20,21,30,31,32,41,42,52,73").split(","))]
    res = {(c, mode): [] for c in combos for mode in ("res", "cold")}
    for r in range(int(os.environ.get("AB_ROUNDS", "4")) + 1):
        for (w, d) in combos:
            os.environ["KLR_Q64_DEPTH"] = str(d); ctx.reload_env()
            for mode in ("res", "cold"):
                ctx.prof_enable(True); ctx.prof_read(reset=True)
                for _ in range(6):
                    if mode == "cold":
                        ctx.evict(1 << 30)
                    K.sweep_select(ctx, mats[w], None, None, np.zeros(1), cw, N)
                nl, ms = ctx.prof_read(reset=True)
                ctx.prof_enable(False)
                if r:
                    res[((w, d), mode)].append(ms / nl * 1e3)
    for c in combos:
        print("%s wgs/cu %d depth %3d  resident %7.2f us  cold %7.2f us" % (sh, c[0], c[1], np.median(res[(c, "res")]), np.median(res[(c, "cold")])), flush=True)
    for w in mats:
        mats[w].close()
