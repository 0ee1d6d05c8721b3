"""Dev tool (GPU, KLR_ABLATIONS build): where the q64 sweep kernel's time goes -- full kernel against its timing-only
ablations (no write-out / no atomics / no S loads / stream only) on the configs[2] and configs[3] shapes."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmerlr_amd as K  # noqa: E402

SHAPES = {"c2": (100000, 65536, 100, "q64c12"), "c3": (125000, 1048576, 500, "q64c13")}
ctx = K.Context(0)
for sh in (sys.argv[1:] or ["c2", "c3"]):
    n, m, N, fmt = SHAPES[sh]
    os.environ["KLR_ENTRY_FORMAT"] = os.environ.get("AB_FMT", fmt); ctx.reload_env()
    rowptr, col, val, labels = K.synth_csr(20261004, 0, n, n, m, 993, 2 * N)
    mat = K.Matrix.from_csr(ctx, n, m + 1, rowptr, col, val, labels)
    del rowptr, col, val
    cw = mat.class_weights()
    os.environ["KLR_STREAM_NT"] = "1"; ctx.reload_env()
    names = {0: "full (nt)", 1: "no write-out", 2: "no atomics", 3: "no S loads", 4: "stream only"}
    res = {a: [] for a in names}
    for r in range(5):
        for a in names:
            os.environ["KLR_Q64_ABL"] = str(a); ctx.reload_env()
            ctx.prof_enable(True); ctx.prof_read(reset=True)
            for _ in range(6):
                ctx.evict(1 << 30)
                K.sweep_select(ctx, mat, None, None, np.zeros(1), cw, N)
            nl, ms = ctx.prof_read(reset=True)
            ctx.prof_enable(False)
            if r:
                res[a].append(ms / nl * 1e3)
    for a in names:
        print("%s %s %-14s cold %7.2f us" % (sh, mat.layout()["format"], names[a], np.median(res[a])), flush=True)
    mat.close()
