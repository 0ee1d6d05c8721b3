"""Dev tool (GPU): the `kmerlr learn` command end to end on synthetic FASTA files of the configs[2] shape (2 x 100,000
sequences of 1,000 nt, 8-mers, --lambda-auto=100): wall time of the whole command and of its phases (-v time stamps).
python tools/cli_end_to_end.py [sequences per class] [extra options ...]"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "kmerlr_amd", "_lib", "kmerlr")
nseq = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 100000
extra = [a for a in sys.argv[1:] if not a.isdigit()]
d = tempfile.mkdtemp(prefix="klr_e2e_")
rng = np.random.default_rng(7)
lut = np.frombuffer(b"acgt", dtype=np.uint8)
t0 = time.perf_counter()
for name, bias in (("fg.fa", 0.27), ("bg.fa", 0.25)):
    p = np.array([bias, (1 - bias) / 3, (1 - bias) / 3, (1 - bias) / 3])
    with open(os.path.join(d, name), "wb") as f:
        for c0 in range(0, nseq, 10000):
            c = min(10000, nseq - c0)
            s = lut[rng.choice(4, size=(c, 1000), p=p)]
            rows = [b">s%d\n" % (c0 + i) + s[i].tobytes() + b"\n" for i in range(c)]
            f.write(b"".join(rows))
print("wrote 2 x %d sequences in %.1f s" % (nseq, time.perf_counter() - t0), flush=True)
for rep in range(2):
    t0 = time.perf_counter()
    r = subprocess.run([CLI, "-v", "-v", "learn", "--lambda-auto=100"] + extra + ["8", "8", "fg.fa", "bg.fa", "out"], cwd=d, capture_output=True, text=True)
    dt = time.perf_counter() - t0
    print("run %d: rc %d, %.3f s wall" % (rep, r.returncode, dt), flush=True)
    if r.returncode != 0:
        print(r.stderr[-2000:])
lines = r.stderr.splitlines()
print("\n".join(ln for ln in lines if ln.startswith("[")))
