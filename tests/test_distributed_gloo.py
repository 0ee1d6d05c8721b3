"""N > 1 on the CPU: world_size-2 gloo run of the product's host driver over row shards.  Each rank holds
half of the rows (n_total = all rows); every sum over samples is all-reduced; both ranks must reach the
same selection, lambda sequence and coefficients as the single-rank run."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from tests import oracle_driver as OD
    from kmerlr_amd.sharding import shard_rows
    n, m = 900, 512
    x, labels = O.synth_csr(99, 0, n, n, m, 120, 12)
    r0, r1 = shard_rows(n, rank, world)
    rp = x.rowptr[r0:r1 + 1] - x.rowptr[r0]
    sl = slice(x.rowptr[r0], x.rowptr[r1])
    xs = O.Csr(r1 - r0, x.dim, rp, x.col[sl], x.val[sl])

    def allreduce(a):
        t = torch.from_numpy(a)
        dist.all_reduce(t)

    OD.set_sharding(n, allreduce)
    h = OD.matrix(xs, labels[r0:r1], n_total=n)
    est = OD.estimator(balance=1, save_trace=1)
    cls = est.estimate(h, [4, 9])
    lams, nz = est.epochs()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), cols0=cls[0]["cols"], th0=cls[0]["theta"], cols1=cls[1]["cols"],
             th1=cls[1]["theta"], lams=np.array(lams), nz=np.array(nz), ntrace=len(est.trace()))
    OD.set_sharding(0, None)
    OD.free_matrix(h)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_driver_matches_single_rank(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    from oracle import oracle as O
    from tests import oracle_driver as OD
    x, labels = O.synth_csr(99, 0, 900, 900, 512, 120, 12)
    h = OD.matrix(x, labels)
    est = OD.estimator(balance=1, save_trace=1)
    cls = est.estimate(h, [4, 9])
    lams, nz = est.epochs()
    OD.free_matrix(h)
    r = [np.load(tmp_path / ("rank%d.npz" % k)) for k in range(world)]
    for k in range(world):
        np.testing.assert_array_equal(r[k]["cols0"], cls[0]["cols"])       # identical selected feature indices
        np.testing.assert_array_equal(r[k]["cols1"], cls[1]["cols"])
        np.testing.assert_array_equal(r[k]["nz"], nz)                      # identical path sequence
        np.testing.assert_allclose(r[k]["lams"], lams, rtol=1e-9)
        np.testing.assert_allclose(r[k]["th0"], cls[0]["theta"], rtol=1e-6, atol=1e-12)
        np.testing.assert_allclose(r[k]["th1"], cls[1]["theta"], rtol=1e-6, atol=1e-12)
    # the ranks agree with each other bit for bit (same all-reduced numbers, same host logic)
    np.testing.assert_array_equal(r[0]["th1"], r[1]["th1"])
    np.testing.assert_array_equal(r[0]["lams"], r[1]["lams"])


def test_shard_rows_partition():
    from kmerlr_amd.sharding import shard_rows
    for n in (0, 1, 7, 100, 100001):
        for w in (1, 2, 3, 8):
            spans = [shard_rows(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
