"""Dev tool (GPU, KLR_ABLATIONS build): wall-clock stamps (100 MHz) of the q64 sweep kernel's phases per workgroup -- where the
fixed ~9 us of a launch go (start skew, first data, drain, write-out).  python tools/q64_phases.py [c2|c3]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmerlr_amd as K  # noqa: E402
from kmerlr_amd import _capi  # noqa: E402

SHAPES = {"c2": (100000, 65536, 100, "q64c12"), "c3": (125000, 1048576, 500, "q64c13")}
ctx = K.Context(0)
lib = _capi.lib()
for sh in (sys.argv[1:] or ["c2"]):
    n, m, N, fmt = SHAPES[sh]
    os.environ["KLR_ENTRY_FORMAT"] = fmt; ctx.reload_env()
    rowptr, col, val, labels = K.synth_csr(20261004, 0, n, n, m, 993, 2 * N)
    mat = K.Matrix.from_csr(ctx, n, m + 1, rowptr, col, val, labels)
    cw = mat.class_weights()
    for mode in ("resident", "cold"):
        for _ in range(3):
            if mode == "cold":
                ctx.evict(1 << 30)
            K.sweep_select(ctx, mat, None, None, np.zeros(1), cw, N)
        buf = np.zeros(2 * 1024 * 8, dtype=np.uint64)
        rc = lib.klr_debug_q64_ts(buf.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0
        nwg = 256
        t = buf.reshape(2, 1024, 8)[:, :nwg, :6].astype(np.int64)
        t0 = t[0, :, 0].min()
        us = (t - t0) / 100.0
        names = ["entry", "2nd piece: L steps done", "1st piece: loop done", "1st piece: after barrier", "1st piece: written out", "kernel: written out"]
        print("%s %s (us after the first workgroup's entry; min / median / max over %d workgroups)" % (sh, mode, nwg))
        for w, wn in ((0, "wave 0 "), (1, "wave 15")):
            for i, nm in enumerate(names):
                v = us[w, :, i]
                print("  %s %-20s %6.2f %6.2f %6.2f" % (wn, nm, v.min(), np.median(v), v.max()))
    mat.close()

# are the slow workgroups the same ones from launch to launch?  (systematic: by XCD = blockIdx % 8, or by position)
for sh in (sys.argv[1:] or ["c2"]):
    n, m, N, fmt = SHAPES[sh]
    os.environ["KLR_ENTRY_FORMAT"] = fmt; ctx.reload_env()
    rowptr, col, val, labels = K.synth_csr(20261004, 0, n, n, m, 993, 2 * N)
    mat = K.Matrix.from_csr(ctx, n, m + 1, rowptr, col, val, labels)
    cw = mat.class_weights()
    durs = []
    for rep in range(6):
        ctx.evict(1 << 30)
        K.sweep_select(ctx, mat, None, None, np.zeros(1), cw, N)
        buf = np.zeros(2 * 1024 * 8, dtype=np.uint64)
        lib.klr_debug_q64_ts(buf.ctypes.data_as(ctypes.c_void_p))
        t = buf.reshape(2, 1024, 8)[0, :256, :].astype(np.int64)
        first_blocks = t[:, 7] >> 8
        t[:, 7] &= 255
        durs.append((t[:, 5] - t[:, 0].min()) / 100.0)
        if rep == 5:
            two = t[:, 7] == 2
            tt = (t[two][:, :6] - t[:, 0].min()) / 100.0
            print("%s crossing workgroups (wave 0), median us: 1st piece loop done %.2f, barrier +%.2f, written out +%.2f, 2nd piece first L steps +%.2f, kernel end +%.2f; blocks %d"
                  % (sh, np.median(tt[:, 2]), np.median(tt[:, 3] - tt[:, 2]), np.median(tt[:, 4] - tt[:, 3]), np.median(tt[:, 1] - tt[:, 4]),
                     np.median(tt[:, 5] - tt[:, 1]), np.median(t[two][:, 6])))
            if os.environ.get("PHASES_CROSS"):
                for i in np.where(two)[0][:40]:
                    r = (t[i, :6] - t[:, 0].min()) / 100.0
                    print("    wg %3d blocks %4d = %4d + %4d | piece 1: %.2f us (%.4f us/block), boundary %.2f, piece 2: %.2f us (%.4f us/block)"
                          % (i, t[i, 6], first_blocks[i], t[i, 6] - first_blocks[i], r[2], r[2] / max(first_blocks[i], 1), r[4] - r[2],
                             r[5] - r[4], (r[5] - r[4]) / max(t[i, 6] - first_blocks[i], 1)))
            one = t[:, 7] == 1
            to = (t[one][:, :6] - t[:, 0].min()) / 100.0
            print("%s single-piece workgroups (wave 0), median us: loop done %.2f, barrier +%.2f, written out +%.2f; blocks %d"
                  % (sh, np.median(to[:, 2]), np.median(to[:, 3] - to[:, 2]), np.median(to[:, 4] - to[:, 3]), np.median(t[one][:, 6])))
        blocks, pieces = t[:, 6].astype(float), t[:, 7].astype(float)
    D = np.array(durs)
    print("%s loop-done time per workgroup over 6 cold launches: mean %.2f, per-launch max %s" % (sh, D.mean(), np.round(D.max(axis=1), 2)))
    print("  correlation of per-workgroup times between launches:", np.round(np.corrcoef(D)[0, 1:], 2))
    mean_wg = D.mean(axis=0)
    A = np.stack([np.ones_like(blocks), blocks, pieces], axis=1)
    coef, *_ = np.linalg.lstsq(A, mean_wg, rcond=None)
    print("  blocks per workgroup min/mean/max %d %.0f %d; pieces per workgroup %s" % (blocks.min(), blocks.mean(), blocks.max(), np.bincount(pieces.astype(int))))
    print("  fit: written-out time = %.2f us + %.4f us/block + %.2f us/piece; residual sd %.2f us" % (coef[0], coef[1], coef[2], (mean_wg - A @ coef).std()))
    print("  by XCD (blockIdx %% 8):", np.round([mean_wg[x::8].mean() for x in range(8)], 2))
    print("  by blockIdx // 32:", np.round([mean_wg[32 * i:32 * i + 32].mean() for i in range(8)], 2))
    if os.environ.get("PHASES_DUMP"):
        for i in range(256):
            print("    wg %3d blocks %5d pieces %d time %6.2f" % (i, blocks[i], pieces[i], mean_wg[i]))
    order = np.argsort(mean_wg)
    print("  slowest 12 workgroups:", order[-12:], np.round(mean_wg[order[-12:]], 2))
    print("  fastest 12 workgroups:", order[:12], np.round(mean_wg[order[:12]], 2))
    mat.close()
