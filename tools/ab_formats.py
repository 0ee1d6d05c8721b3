"""Dev tool (GPU only): the sweep kernel's time per entry format of the packed stream, resident and cold, on the
configs[2] shape and on configs[3]'s per-GPU shard.

  python tools/ab_formats.py [shape ...]      shape = c2 | c3 | rows,cols       (default: c2 c3)
  AB_FORMATS=q64c12,q64c13,r64,u24 AB_ROUNDS=4
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmerlr_amd as K  # noqa: E402

SHAPES = {"c2": (100000, 65536, 100), "c3": (125000, 1048576, 500)}
formats = os.environ.get("AB_FORMATS", "q64c12,q64c13,r64,u24").split(",")
rounds = int(os.environ.get("AB_ROUNDS", "4"))
shapes = sys.argv[1:] or ["c2", "c3"]
ctx = K.Context(0)
for sh in shapes:
    if sh in SHAPES:
        n, m, N = SHAPES[sh]
    else:
        n, m = (int(v) for v in sh.split(","))
        N = 100
    rowptr, col, val, labels = K.synth_csr(20261004, 0, n, n, m, 993, 2 * N)
    mats = {}
    for f in formats:
        f0, _, cutmode = f.partition("%")                                      # q64c13%snap: the old cuts
        os.environ["KLR_Q64_CUT"] = cutmode.split(":")[0] or "comp"; ctx.reload_env()
        os.environ["KLR_Q64_SNAP_TOL"] = cutmode.split(":")[1] if ":" in cutmode else "0.04"; ctx.reload_env()
        base, _, bank = f0.partition("#")                                      # q64c12#0: columns left in ascending order
        os.environ["KLR_ENTRY_FORMAT"] = base.split("@")[0]; ctx.reload_env()
        os.environ["KLR_Q64_WGS"] = base.split("@")[1] if "@" in base else "1"; ctx.reload_env()  # q64c13@2: two workgroups per CU
        os.environ["KLR_Q64_BANK"] = bank or "16"; ctx.reload_env()
        mats[f] = K.Matrix.from_csr(ctx, n, m + 1, rowptr, col, val, labels)
    del rowptr, col, val
    cw = mats[formats[0]].class_weights()
    g0 = None
    res = {f: {"res": [], "cold": []} for f in formats}
    for r in range(rounds + 1):
        for f in formats:
            mt = mats[f]
            for mode in ("res", "cold"):
                ctx.prof_enable(True)
                ctx.prof_read(reset=True)
                for _ in range(6):
                    if mode == "cold":
                        ctx.evict(1 << 30)
                    out = K.sweep_select(ctx, mt, None, None, np.zeros(1), cw, N, want_g=(r == 0 and mode == "res"))
                nl, ms = ctx.prof_read(reset=True)
                ctx.prof_enable(False)
                if r == 0 and mode == "res":
                    if g0 is None:
                        g0 = out["g"]
                    else:
                        assert np.array_equal(g0, out["g"]), "format %s disagrees" % f
                if r > 0:
                    res[f][mode].append(ms / nl * 1e3)
    for f in formats:
        lay = mats[f].layout()
        alg = (3.0 + lay["value_bytes"]) * lay["padded_entries"] + lay["nseg"] * lay["piece_bytes"] + 8.0 * n + lay["table_bytes"]
        a, c = np.median(res[f]["res"]), np.median(res[f]["cold"])
        print("%s %-9s fmt=%-7s stream %7.1f MB alg %7.1f MB pieces %4d | resident %7.2f us %5.3f | cold %7.2f us %5.3f"
              % (sh, f, lay["format"], lay["stream_bytes"] / 1e6, alg / 1e6, lay["nseg"], a, alg / a / 1e3 / 8000.0,
                 c, alg / c / 1e3 / 8000.0), flush=True)
    for f in formats:
        mats[f].close()
