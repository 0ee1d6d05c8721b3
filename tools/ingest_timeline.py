#!/usr/bin/env python3
"""FASTA-shaped input -> resident matrix on the device (klr_mat_from_sequences): k-mer counting, CSR/CSC and the tiled
stream built on the GPU.  python3 tools/ingest_timeline.py [sequences] [length] [k]
Random ACGT sequences of the given length (configs[2] shape: 100000 x 1000, k = 8 -> 993 windows per row)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kmerlr_amd as K                         # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    rng = np.random.default_rng(7)
    t0 = time.perf_counter()
    codes = rng.integers(0, 4, size=(n, L), dtype=np.uint8)
    seqs = [bytes(row) for row in np.frombuffer(b"ACGT", dtype=np.uint8)[codes]]
    labels = (np.arange(n) < n // 2).astype(np.uint8)
    t_gen = time.perf_counter() - t0
    ctx = K.Context(0)
    out = []
    for rep in range(3):
        t0 = time.perf_counter()
        m = K.Matrix.from_sequences(ctx, seqs, labels, k, k)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        out.append(dict(seconds=round(dt, 4), nnz=int(m.nnz), dim=int(m.dim), tiled=m.layout()["tiled"]))
        m.close()
    print(json.dumps(dict(sequences=n, length=L, k=k, host_generate_s=round(t_gen, 2), runs=out)))


if __name__ == "__main__":
    main()
