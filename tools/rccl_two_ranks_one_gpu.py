"""Dev probe (GPU): can RCCL form a 2-rank communicator with both ranks on ONE device here?  (Rounds 1-2 recorded: refused.)
  python tools/rccl_two_ranks_one_gpu.py     -- spawns two processes, 60 s limit each"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import numpy as np
    import kmerlr_amd as K
    rank, path = int(sys.argv[1]), sys.argv[2]
    ctx = K.Context(0)
    if rank == 0:
        uid = K.rccl_unique_id()
        open(path + ".tmp", "wb").write(uid)
        os.rename(path + ".tmp", path)
    else:
        import time
        for _ in range(600):
            if os.path.exists(path):
                break
            time.sleep(0.05)
        uid = open(path, "rb").read()
    try:
        ctx.init_rccl(uid, rank, 2)
        print("rank %d: communicator up, self-check %s" % (rank, ctx.comm_stats()), flush=True)
        a = np.arange(5, dtype=np.int64) + rank
        ctx.allreduce_host(a)
        print("rank %d: allreduce ->" % rank, a, flush=True)
    except Exception as e:
        print("rank %d: FAILED: %s" % (rank, e), flush=True)
    sys.exit(0)
d = tempfile.mkdtemp()
env = dict(os.environ, NCCL_DEBUG="WARN")
ps = [subprocess.Popen(["timeout", "-k", "5", "60", sys.executable, os.path.abspath(__file__), str(r), os.path.join(d, "uid")], env=env) for r in range(2)]
for p in ps:
    p.wait()
print("exit codes", [p.returncode for p in ps])
