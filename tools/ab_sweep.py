"""In-process interleaved A/B timing of the sweep-kernel variants (dev tool; GPU only).
usage: python tools/ab_sweep.py [rounds]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmerlr_amd as K  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2, 3, 4, 5, 6, 7]
n, m, draws, N = int(os.environ.get('AB_N', 100000)), int(os.environ.get('AB_M', 65536)), 993, 100
ctx = K.Context(0)
rowptr, col, val, labels = K.synth_csr(20261004, 0, n, n, m, draws, 2 * N)
mats = {}
for order in ("ct", "rt"):
    os.environ["KLR_TILE_ORDER"] = order; ctx.reload_env()
    mats[order] = K.Matrix.from_csr(ctx, n, m + 1, rowptr, col, val, labels)
    print(order, mats[order].layout())
full = mats["ct"]
cw = full.class_weights()
L = K.lib()
L.klr_debug_set_coo_variant.argtypes = [C.c_int]
res = {(o, v): [] for v in variants for o in mats}
ctx.prof_enable(True)
for r in range(rounds + 1):
    for v in variants:
        for o in mats:
            L.klr_debug_set_coo_variant(v)
            ctx.prof_read(reset=True)
            for _ in range(5):
                K.sweep_select(ctx, mats[o], None, None, np.zeros(1), cw, N)
            nl, ms = ctx.prof_read(reset=True)
            if r > 0:
                res[(o, v)].append(ms / nl * 1e3)
names = {0: "shipped", 1: "512x4 nt", 2: "512x8", 3: "512x8 nt", 4: "1024x4", 5: "1024x4 nt", 6: "1024x2", 7: "256x8", 8: "512 noatom", 9: "512 nogath", 10: "512x2 nt", 11: "1024x2 nt", 12: "1024x1 nt", 13: "512x1 nt", 14: "1024x1", 15: "768x1 nt", 16: "512x4", 21: "noatom", 22: "nogather", 26: "stream", 50: "nomul"}
for v in variants:
    for o in mats:
        a = np.array(res[(o, v)])
        print("order %s variant %d %-10s median %7.2f us  min %7.2f us" % (o, v, names.get(v, "?"), np.median(a), a.min()))
