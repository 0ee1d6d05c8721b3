"""Dev tool (GPU): exactly bench.py's step (theta on the reduced matrix of one solved epoch) and nothing else, for a kernel
trace of the step alone:
  rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/bench_step_trace.py [c2|c3] [steps]
  python tools/trace_view.py OUT/*/*_kernel_trace.csv tl k_row_pass -3 12
Without the profiler it prints the host-timed step."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmerlr_amd as K  # noqa: E402

SHAPES = {"c2": (100000, 65536, 100), "c3": (125000, 1048576, 500)}
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
n, m, N = SHAPES[which]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ctx = K.Context(0)
if os.environ.get("STEPS_COMM"):                       # the library's own RCCL communicator, one rank: the multi-GPU code path
    ctx.init_rccl(K.rccl_unique_id(), 0, 1)
rowptr, col, val, labels = K.synth_csr(20261004, 0, n, n, m, 993, 2 * N)
mat = K.Matrix.from_csr(ctx, n, m + 1, rowptr, col, val, labels)
del rowptr, col, val
cw = mat.class_weights()
r0 = K.sweep_select(ctx, mat, None, None, np.zeros(1), cw, N)
cols = np.nonzero(r0["b"])[0]
red = mat.select_columns(cols)
theta, recs = K.cd_solve(ctx, red, np.zeros(len(cols)), cw=cw, l1reg=r0["lam"] * n, eps_loss=1e-8)
st = K.SweepStep(ctx, mat, red, cols, theta, cw, N, lambda_prev=r0["lam"] * n)
for _ in range(5):
    st.run()
ctx.synchronize()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(steps):
        st.run()
    ctx.synchronize()
    best = min(best, (time.perf_counter() - t0) / steps * 1e3)
ctx.prof_enable(True); ctx.prof_read(reset=True)
for _ in range(steps):
    st.run()
nl, ms = ctx.prof_read(reset=True)
ctx.prof_enable(False)
print("%s: ms per step %.4f (best of 3 x %d); sweep kernel %.2f us; lambda %.6e c %d" % (which, best, steps, ms / max(nl, 1) * 1e3, st.lam.value, st.c.value))
