"""Dev tool (GPU): N full-width steps (sweep + top-N + lambda) and nothing else, for a kernel trace of the step alone:
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d OUT -- python3 tools/steps_only.py [c2|c3] [steps]
  python tools/step_timeline.py OUT"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmerlr_amd as K  # noqa: E402

SHAPES = {"c2": (100000, 65536, 100), "c3": (125000, 1048576, 500)}
n, m, N = SHAPES[sys.argv[1] if len(sys.argv) > 1 else "c2"]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ctx = K.Context(0)
if os.environ.get("STEPS_COMM"):                       # the library's own RCCL communicator, one rank: the multi-GPU code path
    ctx.init_rccl(K.rccl_unique_id(), 0, 1)
rowptr, col, val, labels = K.synth_csr(20261004, 0, n, n, m, 993, 2 * N)
mat = K.Matrix.from_csr(ctx, n, m + 1, rowptr, col, val, labels)
del rowptr, col, val
cw = mat.class_weights()
out = K.sweep_select(ctx, mat, None, None, np.zeros(1), cw, N)
idx = np.flatnonzero(out["b"]).astype(np.int64) if "b" in out else None
for _ in range(5):
    K.sweep_select(ctx, mat, None, None, np.zeros(1), cw, N)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    K.sweep_select(ctx, mat, None, None, np.zeros(1), cw, N)
ctx.synchronize()
print("ms per step %.4f" % ((time.perf_counter() - t0) / steps * 1e3))
