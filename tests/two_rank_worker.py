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
    if os.environ.get("KLR_TWO_RANK_MODE") == "scatter":
        return scatter_collective(K, dist, ctx, hip, rank, world, calls)
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


def scatter_collective(K, dist, ctx, hip, rank, world, calls):
    """The sweep's collective as reduce-scatter + local top-2N + all-gather of the candidates (klr_ctx_set_collectives)
    against the all-reduce form on the same shards: selection, lambda and loss must be BITWISE the same; whole leapfrog
    runs through the driver must give identical classifiers."""
    import torch
    from kmerlr_amd import _driver
    n_total, m, draws, N = 12000, 40000, 300, 25
    n_local = n_total // world
    cnt = {"rs": 0, "ag": 0}

    def d2h(ptr, nbytes):
        host = np.empty(nbytes, dtype=np.uint8)
        assert hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(nbytes), 2) == 0
        return host

    def h2d(ptr, host):
        assert hip.hipMemcpy(C.c_void_p(ptr), host.ctypes.data_as(C.c_void_p), C.c_size_t(host.nbytes), 1) == 0

    def rs_hook(send, recv, count, dtype, stream):
        hip.hipStreamSynchronize(C.c_void_p(stream))
        full = d2h(send, count * world * 8).view(np.int64 if dtype == 1 else np.float64).copy()
        t = torch.from_numpy(full)
        dist.all_reduce(t)                                            # gloo has no reduce-scatter: all-reduce, keep the own slice
        h2d(recv, np.ascontiguousarray(full[rank * count:(rank + 1) * count]).view(np.uint8))
        cnt["rs"] += 1

    def ag_hook(send, recv, nbytes, stream):
        hip.hipStreamSynchronize(C.c_void_p(stream))
        mine = torch.from_numpy(d2h(send, nbytes).copy())
        outs = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(outs, mine)
        h2d(recv, np.concatenate([o.numpy() for o in outs]))
        cnt["ag"] += 1

    rowptr, col, val, labels = K.synth_csr(20261004, rank * n_local, n_local, n_total, m, draws, 2 * N)
    shard = K.Matrix.from_csr(ctx, n_local, m + 1, rowptr, col, val, labels, n_total)
    assert shard.layout()["tiled"]
    cw = shard.class_weights()
    # all-reduce form first (no extra hooks installed)
    a0 = K.sweep_select(ctx, shard, None, None, np.zeros(1), cw, N)
    cols = np.nonzero(a0["b"])[0]
    red = shard.select_columns(cols)
    theta, _ = K.cd_solve(ctx, red, np.zeros(len(cols)), cw=cw, l1reg=a0["lam"] * n_total, eps_loss=1e-10)
    a1 = K.sweep_select(ctx, shard, red, cols, theta, cw, N)
    est = _driver.hip_estimator(ctx)
    ca = est.estimate(shard._h, [10, N]); ea = est.epochs(); est.close()
    at = K.sweep_select(ctx, shard, None, None, np.zeros(1), (1.0, 1.0), 400)     # tied count data, wide selection
    # reduce-scatter form: used only once the hooks passed the collective self-check
    ctx.set_collectives(rs_hook, ag_hook)
    u0 = K.sweep_select(ctx, shard, None, None, np.zeros(1), cw, N)
    assert cnt["rs"] == 0 and not ctx.comm_stats()["checked"], "unchecked hooks must not carry the sweep"
    ctx.comm_selfcheck()
    assert ctx.comm_stats()["checked"] and cnt == {"rs": 1, "ag": 1}, cnt
    ctx.set_option("KLR_SWEEP_RANGES", 1)                              # one reduce-scatter per sweep: the counts below
    ctx.set_option("KLR_RS", 1)                                        # ... and no timing probe of the hooks (these serve the
                                                                       # reduce-scatter through an all-reduce: the probe would choose that)
    i64_before = calls["i64"]
    s0 = K.sweep_select(ctx, shard, None, None, np.zeros(1), cw, N)
    # the first reduce-scatter sweep on a matrix is repeated as an all-reduce and compared (one int64 all-reduce) ...
    assert cnt == {"rs": 2, "ag": 2} and calls["i64"] == i64_before + 1, (cnt, calls)
    s1 = K.sweep_select(ctx, shard, red, cols, theta, cw, N)
    # ... later ones post no all-reduce at all (a hint from the previous sweep that proves too high after the model moved
    # costs a second local selection and a second all-gather, not a second reduction)
    assert cnt["rs"] == 3 and cnt["ag"] in (3, 4) and calls["i64"] == i64_before + 1, (cnt, calls)
    assert u0["lam"] == s0["lam"]
    for a, s in ((a0, s0), (a1, s1)):
        np.testing.assert_array_equal(a["b"], s["b"])
        assert a["lam"] == s["lam"] and a["loss"] == s["loss"] and a["c"] == s["c"] and a["ok"] == s["ok"]
    est = _driver.hip_estimator(ctx)
    cs = est.estimate(shard._h, [10, N]); es = est.epochs(); est.close()
    assert es == ea
    for a, s in zip(ca, cs):
        np.testing.assert_array_equal(a["cols"], s["cols"]); np.testing.assert_array_equal(a["theta"], s["theta"])
    # tied count data at theta = 0 with a selection that straddles tie groups: the same result, through the candidates or
    # -- when a rank's candidate list cannot prove the largest |g| below the N-th -- through the all-reduce fallback
    st = K.sweep_select(ctx, shard, None, None, np.zeros(1), (1.0, 1.0), 400)
    np.testing.assert_array_equal(at["b"], st["b"])
    assert at["lam"] == st["lam"] and at["loss"] == st["loss"] and at["c"] == st["c"]
    # strip ranges of the overlapped form (one sweep launch per range, finish + reduce-scatter of range g on a second stream
    # while range g + 1 streams): whatever the number of ranges, the same integers are summed -- bitwise the same result
    for G in (1, 2, 3, 5):
        ctx.set_option("KLR_SWEEP_RANGES", G)
        g0 = K.sweep_select(ctx, shard, None, None, np.zeros(1), cw, N)
        g1 = K.sweep_select(ctx, shard, red, cols, theta, cw, N)
        gt = K.sweep_select(ctx, shard, None, None, np.zeros(1), (1.0, 1.0), 400)
        for a, s_ in ((a0, g0), (a1, g1), (at, gt)):
            np.testing.assert_array_equal(a["b"], s_["b"])
            assert a["lam"] == s_["lam"] and a["loss"] == s_["loss"] and a["c"] == s_["c"] and a["ok"] == s_["ok"], (G, a["lam"], s_["lam"])
    ctx.set_option("KLR_SWEEP_RANGES", None)
    ctx.set_option("KLR_RS", None)
    # left to itself the context times the hooks and decides: every rank reports the same choice, and the result is the same
    p0 = K.sweep_select(ctx, shard, None, None, np.zeros(1), cw, N)
    pr = ctx.comm_probe()
    assert pr["elements"] == m + 6 and pr["form"] in ("reduce-scatter + all-gather", "all-reduce") and pr["ranges"] in (1, 2), pr
    both = np.zeros(world * 2); both[2 * rank] = 1.0 if pr["form"].startswith("reduce") else 0.0; both[2 * rank + 1] = pr["ranges"]
    ctx.allreduce_host(both)
    assert both[0] == both[2] and both[1] == both[3], both
    np.testing.assert_array_equal(a0["b"], p0["b"]); assert a0["lam"] == p0["lam"]
    st_ = ctx.comm_stats()
    assert st_["sweeps_scatter"] >= 3 and st_["sweeps_allreduce"] >= 3, st_
    if rank == 0:
        print("two ranks ok (scatter): lambda=%e rs=%d ag=%d fallbacks=%d" % (s0["lam"], cnt["rs"], cnt["ag"], st_["scatter_fallbacks"]))
    dist.barrier()
    dist.destroy_process_group()
    return 0


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
