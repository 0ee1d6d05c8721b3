"""Dev tool (GPU, KLR_ABLATIONS build): per-workgroup {entry, end} stamps of k_sweep_q64 over many launches, saved for
offline analysis (which workgroups end last, how persistent that is).  python tools/q64_wg_dump.py out.npz [c2|c3] [launches]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmerlr_amd as K  # noqa: E402
from kmerlr_amd import _capi  # noqa: E402

SHAPES = {"c2": (100000, 65536, 100, "q64c12"), "c3": (125000, 1048576, 500, "q64c13")}
out = sys.argv[1]
shapes = [s for s in sys.argv[2:] if s in SHAPES] or ["c2"]
reps = int([s for s in sys.argv[2:] if s.isdigit()][0]) if [s for s in sys.argv[2:] if s.isdigit()] else 30
ctx = K.Context(0)
lib = _capi.lib()
res = {}
for sh in shapes:
    n, m, N, fmt = SHAPES[sh]
    os.environ["KLR_ENTRY_FORMAT"] = fmt; ctx.reload_env()
    rowptr, col, val, labels = K.synth_csr(20261004, 0, n, n, m, 993, 2 * N)
    for inst in range(2):                       # two instances of the same matrix: does the pattern follow the allocation?
        mat = K.Matrix.from_csr(ctx, n, m + 1, rowptr, col, val, labels)
        cw = mat.class_weights()
        for mode in ("resident", "cold"):
            T = []
            for r in range(reps + 3):
                if mode == "cold":
                    ctx.evict(1 << 30)
                K.sweep_select(ctx, mat, None, None, np.zeros(1), cw, N)
                buf = np.zeros(2 * 1024 * 8, dtype=np.uint64)
                assert lib.klr_debug_q64_ts(buf.ctypes.data_as(ctypes.c_void_p)) == 0
                if r >= 3:
                    T.append(buf.reshape(2, 1024, 8)[:, :256, :].astype(np.int64))
            res["%s_%d_%s" % (sh, inst, mode)] = np.array(T)
        mat.close()
np.savez_compressed(out, **res)
print("saved", out, list(res))
