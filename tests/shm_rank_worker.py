"""Worker of tests/test_gpu_shm_ranks.py: one of N processes sharing cuda:0, the library's built-in shared-memory
collectives (klr_ctx_init_shm) between them.  argv[1] = case; RANK / WORLD_SIZE / KLR_SHM_NAME from the environment."""
import ctypes as C
import os
import sys
from multiprocessing import shared_memory

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def dense_rows(seed, n, d, n_total_rows, row0):
    """rows [row0, row0 + n) of a dense score table with a planted linear signal; reference layout (bias first)."""
    rng = np.random.default_rng(seed)
    x = rng.normal(size=(n_total_rows, d - 1))
    w = np.zeros(d - 1); w[:4] = [1.5, -1.0, 0.8, -0.6]
    y = (x @ w + 0.3 * rng.normal(size=n_total_rows) > 0).astype(np.uint8)
    xs, ys = x[row0:row0 + n], y[row0:row0 + n]
    return xs, ys


def csr_of(xs, drop=None):
    """dense rows -> CSR with every column stored (zeros included), except the entries in `drop` = {(row, column)}"""
    n, m = xs.shape
    rp = [0]; col = []; val = []
    for i in range(n):
        col.append(0); val.append(1.0)
        for j in range(m):
            if drop and (i, j + 1) in drop:
                continue
            col.append(j + 1); val.append(float(xs[i, j]))
        rp.append(len(col))
    return np.array(rp, dtype=np.int64), np.array(col, dtype=np.int64), np.array(val)


def main():
    import kmerlr_amd as K
    case = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    shm = shared_memory.SharedMemory(name=os.environ["KLR_SHM_NAME"])
    from multiprocessing import resource_tracker
    resource_tracker.unregister(shm._name, "shared_memory")          # the creating test unlinks it (Python 3.10 would do it here, too)
    addr = C.addressof(C.c_char.from_buffer(shm.buf))
    ctx = K.Context(0)
    ctx.init_shm(addr, shm.size, rank, world)
    assert ctx.comm_stats()["checked"]
    if case == "dense":
        # the dense-panel MFMA Gram must be chosen by ALL ranks or by none.  Shard 0 is a dense panel;
        # shard 1 lacks ONE entry (and in the second round is dense too): every round must reproduce the one-rank solve.
        n_total, d = 1200, 25
        n_local = n_total // world
        for rnd, drop1 in enumerate(({(5, 7)}, None)):
            xs, ys = dense_rows(4, n_local, d, n_total, rank * n_local)
            drop = drop1 if rank == 1 else None
            rp, col, val = csr_of(xs, drop)
            shard = K.Matrix.from_csr(ctx, n_local, d, rp, col, val, ys, n_total)
            red = shard.select_columns(np.arange(d))
            assert red.nnz == n_local * d - (1 if drop else 0)
            cw = shard.class_weights()
            theta, recs = K.cd_solve(ctx, red, np.zeros(d), cw=cw, l1reg=0.01 * n_total, eps_loss=1e-10)
            out = np.zeros(d + 1); out[:d] = theta; out[d] = len(recs)
            both = np.zeros((world, d + 1)); both[rank] = out
            ctx.allreduce_host(both.reshape(-1))
            np.testing.assert_array_equal(both[0], both[rank])          # replicated solve: every rank holds the same iterate
            if rank == 0:
                base = K.Context(0)
                xa, ya = dense_rows(4, n_total, d, n_total, 0)
                rpa, cola, vala = csr_of(xa, {(n_local + 5, 7)} if drop1 else None)
                full = K.Matrix.from_csr(base, n_total, d, rpa, cola, vala, ya)
                t1, r1 = K.cd_solve(base, full.select_columns(np.arange(d)), np.zeros(d), cw=full.class_weights(), l1reg=0.01 * n_total, eps_loss=1e-10)
                assert len(r1) == len(recs), (len(r1), len(recs))
                np.testing.assert_allclose(theta, t1, rtol=1e-8, atol=1e-12)
                print("dense round %d ok: sweeps=%d nonzero=%d" % (rnd, len(recs), int(np.count_nonzero(theta[1:]))))
                base.close()
            red.close(); shard.close()
    elif case == "collectives":
        # the three hooks on awkward sizes (several slots' worth, odd counts), against numpy
        rng = np.random.default_rng(100 + rank)
        for count in (1, 7, 300001, 524289):
            a = rng.integers(-2 ** 40, 2 ** 40, size=count, dtype=np.int64)
            allv = np.zeros((world, count), dtype=np.int64); allv[rank] = a
            s = a.copy(); ctx.allreduce_host(s)
            ctx.allreduce_host(allv.reshape(-1))
            np.testing.assert_array_equal(s, allv.sum(axis=0))
            f = rng.normal(size=count)
            allf = np.zeros((world, count)); allf[rank] = f
            sf = f.copy(); ctx.allreduce_host(sf)
            ctx.allreduce_host(allf.reshape(-1))
            exp = allf[0].copy()
            for r in range(1, world):
                exp += allf[r]                                         # rank order
            np.testing.assert_array_equal(sf, exp)
        if rank == 0:
            print("collectives ok")
    ctx.close()
    shm.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
