"""Worker of tests/test_gpu_two_ranks.py: one of two processes sharing cuda:0, each holding half of the rows; the
library's all-reduce hook is served by torch.distributed (gloo) through host memory.  Rank 0 also computes the
single-process result on the whole matrix and compares."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    import kmerlr_amd as K
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hip = C.CDLL("libamdhip64.so")
    n_total, m, draws, N = 12000, 4096, 300, 12
    n_local = n_total // world
    calls = {"i64": 0, "f64": 0}

    def hook(buf, count, dtype, stream):
        hip.hipStreamSynchronize(C.c_void_p(stream))
        host = np.empty(count, dtype=np.int64 if dtype == 1 else np.float64)
        assert hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), C.c_void_p(buf), C.c_size_t(count * 8), 2) == 0
        t = torch.from_numpy(host)
        dist.all_reduce(t)
        assert hip.hipMemcpy(C.c_void_p(buf), host.ctypes.data_as(C.c_void_p), C.c_size_t(count * 8), 1) == 0
        calls["i64" if dtype == 1 else "f64"] += 1

    ctx = K.Context(0)
    ctx.set_allreduce(hook, rank, world)
    if os.environ.get("KLR_TWO_RANK_MODE") == "mixed":
        return mixed_layouts(K, dist, ctx, rank, world, calls)
    rowptr, col, val, labels = K.synth_csr(20261004, rank * n_local, n_local, n_total, m, draws, 2 * N)
    shard = K.Matrix.from_csr(ctx, n_local, m + 1, rowptr, col, val, labels, n_total)
    assert shard.layout()["tiled"]
    cw = shard.class_weights()                                   # all-reduced class counts
    r = K.sweep_select(ctx, shard, None, None, np.zeros(1), cw, N, want_g=True)
    cols = np.nonzero(r["b"])[0]
    red = shard.select_columns(cols)
    theta, recs = K.cd_solve(ctx, red, np.zeros(len(cols)), cw=cw, l1reg=r["lam"] * n_total, eps_loss=1e-10)
    r2 = K.sweep_select(ctx, shard, red, cols, theta, cw, N, want_g=True)        # second epoch: theta on the reduced shard
    loss = K.loss(ctx, red, theta, cw=cw, lam=r["lam"])
    assert calls["i64"] >= 2 and calls["f64"] >= 2
    ok = True
    if rank == 0:
        base = K.Context(0)
        rp, cc, vv, ll = K.synth_csr(20261004, 0, n_total, n_total, m, draws, 2 * N)
        full = K.Matrix.from_csr(base, n_total, m + 1, rp, cc, vv, ll)
        cw1 = full.class_weights()
        s = K.sweep_select(base, full, None, None, np.zeros(1), cw1, N, want_g=True)
        np.testing.assert_array_equal(cw, cw1)
        np.testing.assert_array_equal(r["g"], s["g"])             # exact integer sweep: bitwise independent of the sharding
        np.testing.assert_array_equal(r["b"], s["b"]); assert r["lam"] == s["lam"] and r["loss"] == s["loss"]
        fred = full.select_columns(cols)
        t1, recs1 = K.cd_solve(base, fred, np.zeros(len(cols)), cw=cw1, l1reg=s["lam"] * n_total, eps_loss=1e-10)
        assert len(recs) == len(recs1)
        np.testing.assert_allclose(theta, t1, rtol=1e-9, atol=1e-13)
        s2 = K.sweep_select(base, full, fred, cols, t1, cw1, N, want_g=True)
        np.testing.assert_allclose(r2["g"], s2["g"], rtol=0, atol=1e-12 * np.max(np.abs(s2["g"])))
        np.testing.assert_array_equal(r2["b"], s2["b"])
        assert abs(loss - K.loss(base, fred, t1, cw=cw1, lam=s["lam"])) < 1e-12
        print("two ranks ok: lambda=%e sweeps=%d" % (r["lam"], len(recs)))
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 1


def mixed_layouts(K, dist, ctx, rank, world, calls):
    """Shards that fall on different sides of the 'build the integer stream' test must still post matching
    collectives: rank 0 holds 10,400 count rows (tiled stream), rank 1 holds 1,600 rows (below the 2,048-row floor:
    plain CSC only) one of which carries a non-count value.  The first sweep negotiates the protocol (one 2-element
    f64 all-reduce); both ranks then run the plain sweep with the f64 all-reduce.  Rank 0 checks against the
    single-process result on the same rows."""
    n_total, m, draws, N = 12000, 4096, 300, 12
    r0 = 0 if rank == 0 else 10400
    n_local = 10400 if rank == 0 else 1600
    rowptr, col, val, labels = K.synth_csr(20261004, r0, n_local, n_total, m, draws, 2 * N)
    if rank == 1:
        val[int(rowptr[7]) + 3] = 2.5                                 # row 10407 of the data set: not a count
    shard = K.Matrix.from_csr(ctx, n_local, m + 1, rowptr, col, val, labels, n_total)
    assert shard.layout()["tiled"] == (rank == 0), shard.layout()
    cw = shard.class_weights()
    r = K.sweep_select(ctx, shard, None, None, np.zeros(1), cw, N, want_g=True)
    assert calls["i64"] == 0, "a rank posted the int64 all-reduce against a neighbour without the stream"
    cols = np.nonzero(r["b"])[0]
    red = shard.select_columns(cols)
    theta, recs = K.cd_solve(ctx, red, np.zeros(len(cols)), cw=cw, l1reg=r["lam"] * n_total, eps_loss=1e-10)
    r2 = K.sweep_select(ctx, shard, red, cols, theta, cw, N, want_g=True)
    g3 = K.gradient(ctx, shard, cols, theta, cw)                      # klr_gradient goes through the same negotiation
    np.testing.assert_allclose(g3, r2["g"], rtol=0, atol=1e-13 * np.max(np.abs(r2["g"])))
    if rank == 0:
        base = K.Context(0)
        rp, cc, vv, ll = K.synth_csr(20261004, 0, n_total, n_total, m, draws, 2 * N)
        vv[int(rp[10407]) + 3] = 2.5
        full = K.Matrix.from_csr(base, n_total, m + 1, rp, cc, vv, ll)
        assert not full.layout()["tiled"]
        cw1 = full.class_weights()
        s = K.sweep_select(base, full, None, None, np.zeros(1), cw1, N, want_g=True)
        np.testing.assert_allclose(r["g"], s["g"], rtol=0, atol=1e-13 * np.max(np.abs(s["g"])))
        np.testing.assert_array_equal(r["b"], s["b"])
        assert abs(r["lam"] - s["lam"]) <= 1e-12 * s["lam"]
        fred = full.select_columns(cols)
        t1, recs1 = K.cd_solve(base, fred, np.zeros(len(cols)), cw=cw1, l1reg=s["lam"] * n_total, eps_loss=1e-10)
        np.testing.assert_allclose(theta, t1, rtol=1e-7, atol=1e-12)
        s2 = K.sweep_select(base, full, fred, cols, t1, cw1, N, want_g=True)
        np.testing.assert_allclose(r2["g"], s2["g"], rtol=0, atol=1e-9 * np.max(np.abs(s2["g"])))
        print("two ranks ok (mixed layouts): lambda=%e" % r["lam"])
    dist.barrier()
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
