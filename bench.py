#!/usr/bin/env python3
"""bench.py -- full-width coordinate sweep of kmerLr's L1-LR hot path on MI355X.

One "step" = one leapfrog epoch's full-width pass (klr_sweep_select): X.theta / sigmoid / log-loss
over the rows (from the reduced matrix of the current theta), the per-feature gradient sweep
g = X'(cw (sigma - y))/n over the packed k-mer count stream, ONE all-reduce of [g | loss] across ranks,
and the top-2N |g| selection + lambda rule.  Inputs are resident in HBM before the timed region.

Workload (BASELINE.json configs[2]): synthetic 100k sequences x 65,536 8-mers per GPU (weak
scaling: rows grow with N), --lambda-auto=100, theta on 100 selected columns.  At every N the line also
carries configs[3]'s per-GPU shard: 125,000 x 1,048,576, --lambda-auto=500 (--no-configs3 skips it).

  python bench.py [--gpus N] [--steps K] [--warmup W]

With --gpus N > 1 and no WORLD_SIZE in the environment this process only LAUNCHES: it starts
`python -m torch.distributed.run --nproc-per-node N ... bench.py ...` as a child before anything here
touches the GPU, relays its output (rank 0 prints the one JSON line) and exits with its return code.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 20261004
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
PROF_STRIDE = 4            # launches of the sweep kernel per HIP-event pair inside the timed region
EVICT_BYTES = 1 << 30      # written + copied between the steps of the cold-stream leg (Infinity Cache: 256 MiB)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows-per-gpu", type=int, default=100000)
    ap.add_argument("--features", type=int, default=65536)
    ap.add_argument("--draws", type=int, default=993)
    ap.add_argument("--lambda-auto", type=int, default=100)
    ap.add_argument("--cpu-baseline-rows", type=int, default=20000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cd-sweeps", type=int, default=64, help="reduced-problem CD sweeps timed for cd_sweeps_per_s")
    ap.add_argument("--force-comm", action="store_true",
                    help="set up the process group and the library's RCCL communicator even for one rank (plumbing check)")
    ap.add_argument("--configs3", action="store_true", help="(default since round 3) kept for old command lines")
    ap.add_argument("--no-configs3", action="store_true",
                    help="skip BASELINE configs[3]'s per-GPU shard (125,000 x 1,048,576, --lambda-auto=500), which every N runs by default")
    ap.add_argument("--transport", choices=("rccl", "shm"), default="rccl",
                    help="collectives of an N > 1 run: RCCL over xGMI (one GPU per rank), or the library's host-staged shared-memory "
                         "hooks with every rank on the SAME card (rehearsal of the N > 1 code path on a one-GPU box; the line says so)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous rehearsal on the CPU (gloo): no GPU work, the line says dry_run")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_children(args, argv):
    """N > 1 without a launcher: start one, as a child.  Nothing in this process has touched HIP or torch.cuda."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


def dry_run(args, world, rank):
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    dt = 1e-3 * (rank + 1)
    if world > 1:
        dist.barrier()
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({"metric": "samples_x_features_per_s", "value": 0.0, "unit": "samples*features/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic", "dry_run": True,
                          "config": {"workload": "launcher rehearsal (gloo, no GPU work)"}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


class Rank:
    """One process = one GPU: context, communicator, timing helpers."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        import kmerlr_amd as K
        self.torch, self.dist, self.K, self.args = torch, dist, K, args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.shm = args.transport == "shm" and self.world > 1
        if self.shm:
            self.local_rank = 0                    # rehearsal: every rank on the same card
        torch.cuda.set_device(self.local_rank)
        self.comm = self.world > 1 or args.force_comm
        self.tdev = "cpu" if self.shm else "cuda"
        if self.comm:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            import datetime
            if self.shm:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world, timeout=datetime.timedelta(seconds=300))
            else:
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world,
                                        device_id=torch.device("cuda", self.local_rank),
                                        timeout=datetime.timedelta(seconds=300))      # a rank that never arrives fails the run, not hangs it
        self.ctx = K.Context(self.local_rank)
        if self.shm:
            # one shared region for all ranks: rank 0 creates and initialises it, the name travels through the process group
            import ctypes
            from multiprocessing import shared_memory, resource_tracker
            nbytes = K.lib().klr_shm_region_bytes(self.world)
            if self.rank == 0:
                self._shm = shared_memory.SharedMemory(create=True, size=nbytes)
                addr = ctypes.addressof(ctypes.c_char.from_buffer(self._shm.buf))
                assert K.lib().klr_shm_region_init(ctypes.c_void_p(addr), self._shm.size, self.world) == 0
            names = [self._shm.name if self.rank == 0 else None]
            dist.broadcast_object_list(names, src=0)
            if self.rank != 0:
                self._shm = shared_memory.SharedMemory(name=names[0])
                resource_tracker.unregister(self._shm._name, "shared_memory")
                addr = ctypes.addressof(ctypes.c_char.from_buffer(self._shm.buf))
            self._shm_addr = addr
            self.ctx.init_shm(addr, self._shm.size, self.rank, self.world)
            assert self.ctx.comm_stats()["checked"]
        elif self.comm:
            # RCCL communicator owned by the library: one collective per sweep on the library's stream.  klr_ctx_init_rccl
            # ends with the collective self-check ([1, rank] as int64 must come back as [N, N(N-1)/2]; f64, reduce-scatter
            # and all-gather alike) and raises if the communicator does not reduce over the ranks it claims
            ids = [K.rccl_unique_id() if self.rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            self.ctx.init_rccl(ids[0], self.rank, self.world)
            assert self.ctx.comm_stats()["checked"]

    def fence(self):
        self.ctx.synchronize()
        self.torch.cuda.synchronize()
        if self.comm:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def max_over_ranks(self, dt):
        if self.world > 1:
            t = self.torch.tensor([dt], dtype=self.torch.float64, device=self.tdev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def timed_steps(self, step, warmup, steps):
        """W untimed steps, then exactly K steps between two fences; HIP-event time of the sweep kernel inside."""
        self.ctx.prof_reserve(4 * steps + 16)      # the timed steps take their HIP events from the context's pool: none is created
        for _ in range(warmup):
            step()
        self.fence()
        # inside the timed region every PROF_STRIDE-th launch of the sweep kernel carries HIP events (a timed dispatch costs a
        # completion signal, ~8 us of a 75 us step); the cold and in-estimate legs time every launch
        self.ctx.prof_enable(PROF_STRIDE)
        self.ctx.prof_read(reset=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        self.fence()
        dt = time.perf_counter() - t0
        launches, kern_ms = self.ctx.prof_read(reset=True)
        self.ctx.prof_enable(False)
        return self.max_over_ranks(dt), launches, kern_ms

    def cold_kernel_ms(self, step, steps):
        """The same steps with the packed stream evicted from the Infinity Cache before each one (1 GiB written and
        copied on the library's stream); only the sweep kernel's own HIP events are read."""
        self.ctx.prof_enable(True)
        self.ctx.prof_read(reset=True)
        for _ in range(steps):
            self.ctx.evict(EVICT_BYTES)
            step()
        self.ctx.synchronize()
        launches, kern_ms = self.ctx.prof_read(reset=True)
        self.ctx.prof_enable(False)
        return kern_ms / max(launches, 1)

    def close(self):
        self.ctx.close()
        if self.comm:
            self.dist.barrier()
        if self.shm:
            del self._shm_addr
            try:
                self._shm.close()
                if self.rank == 0:
                    self._shm.unlink()
            except Exception:
                pass
        if self.comm:
            self.dist.destroy_process_group()


def alg_bytes_of(lay, n_local, nnz, m):
    """ALGORITHMIC bytes of one launch of the sweep kernel for the layout in HBM."""
    if lay["tiled"]:
        # one pass over the packed stream (2, 3 or 4 bytes per padded unit entry) + one int64 vector per (strip, workgroup)
        # piece written + s read once (re-reads of s tiles are cache hits and not counted) + side tables
        return (3.0 + lay["value_bytes"]) * lay["padded_entries"] + lay["nseg"] * lay["piece_bytes"] + 8.0 * n_local + lay["table_bytes"]
    # plain CSC: 4 B row id + 4|8 B value per stored nonzero, chunk table, s once per row, g once per column
    return (4.0 + lay["value_bytes"]) * nnz + 12.0 * lay["nseg"] + 8.0 * n_local + 8.0 * (m + 1)


def roofline_of(lay, alg_bytes, kern_avg_ms, launches, cold_avg_ms, n_local, nnz, m, traffic=None, traffic_source=None):
    def rate(ms):
        return alg_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    two_pass_bytes = 16.0 * nnz + 8.0 * (n_local + 1) + 8.0 * (m + 1) + 9.0 * n_local + 8.0 * (m + 1)
    achieved = rate(kern_avg_ms)
    r = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
         "kernel": lay["kernel"], "layout": lay, "kernel_avg_ms": round(kern_avg_ms, 4), "launches": launches,
         "launches_note": "launches timed with HIP events inside the timed region: every %d-th of its steps" % PROF_STRIDE,
         "algorithmic_bytes_per_launch": alg_bytes,
         "resident": {"kernel_avg_ms": round(kern_avg_ms, 4), "achieved": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBS, 4),
                      "note": "the timed region: back-to-back steps, a stream that fits the 256 MiB Infinity Cache stays in it"},
         "survey_two_pass_bytes": two_pass_bytes,
         "survey_two_pass_equiv_gbs": round(two_pass_bytes / (kern_avg_ms * 1e-3) / 1e9, 1) if kern_avg_ms > 0 else 0.0}
    if cold_avg_ms is not None:
        r["cold"] = {"kernel_avg_ms": round(cold_avg_ms, 4), "achieved": round(rate(cold_avg_ms), 1),
                     "frac": round(rate(cold_avg_ms) / HBM_PEAK_GBS, 4),
                     "note": "same steps, 1 GiB written and copied before each one: the stream comes from HBM"}
        r["frac_cold"] = r["cold"]["frac"]
    return r


def load_traffic(lay, nnz, m):
    """HBM traffic of the sweep kernel from the committed PMC passes (profiles/): valid for this exact workload AND this
    exact kernel build only -- the file records the hash of the sweep / ingest sources it was collected with (klr_version),
    and numbers of another build are refused (traffic null, the reason in traffic_source)."""
    import kmerlr_amd as K
    have = K.kernel_hash()
    stale = None
    for name in ("r03_pmc_traffic.json", "r03_pmc_traffic_c3.json"):
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", name)))
            w = pj["workload"]
            if (lay["tiled"] and w["nnz"] == nnz and w["features"] == m and w.get("format") == lay.get("format", w.get("format"))
                    and w.get("entry_bytes") == 3 + lay["value_bytes"]):
                if pj.get("kernel_hash") != have:
                    stale = "profiles/%s was collected with kernels:%s, this library is kernels:%s: not used" % (name, pj.get("kernel_hash"), have)
                    continue
                k = pj["kernels"][pj.get("kernel", "k_sweep_coo")]
                return k["hbm_traffic_bytes_per_launch"], "profiles/%s (%s; kernels:%s, collected at commit %s)" % (
                    name, pj.get("how", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes"), pj.get("kernel_hash"), pj.get("collected_at_commit", "?"))
        except Exception:
            continue
    return None, stale


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_children(args, argv))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    args.gpus = world
    if args.dry_run:
        return dry_run(args, world, rank)

    R = Rank(args)
    K, ctx = R.K, R.ctx
    n_local, m, draws, N = args.rows_per_gpu, args.features, args.draws, args.lambda_auto
    n_total = n_local * world
    S = 2 * N

    # ---- synthetic shard, generated on the host and moved to HBM once -------------------------------
    t0 = time.time()
    rowptr, col, val, labels = K.synth_csr(SEED, rank * n_local, n_local, n_total, m, draws, S)
    t_gen = time.time() - t0
    t0 = time.time()
    full = K.Matrix.from_csr(ctx, n_local, m + 1, rowptr, col, val, labels, n_total)
    ctx.synchronize()
    t_up = time.time() - t0
    del rowptr, col, val
    nnz = full.nnz
    cw = full.class_weights()

    # ---- a realistic mid-path state: epoch 0 selection, reduced matrix, solved theta ------------------
    r0 = K.sweep_select(ctx, full, None, None, np.zeros(1), cw, N)
    cols = np.nonzero(r0["b"])[0]
    red = full.select_columns(cols)
    t0 = time.time()
    theta, recs = K.cd_solve(ctx, red, np.zeros(len(cols)), cw=cw, l1reg=r0["lam"] * n_total, eps_loss=1e-8)
    t_solve = time.time() - t0
    n_sweeps_solve = len(recs)

    stepper = K.SweepStep(ctx, full, red, cols, theta, cw, N, lambda_prev=r0["lam"] * n_total)
    dt, launches, kern_ms = R.timed_steps(stepper.run, args.warmup, args.steps)
    out = stepper.result()
    cold_ms = R.cold_kernel_ms(stepper.run, min(args.steps, 20))

    # ---- reduced-problem CD sweep rate (sweeps/s), from the device timestamps of the sweep kernels themselves
    # (klr_cd_solve's own HIP events: soft-threshold sweeps only, no loss pass, no IRLS set-up, no host turnaround)
    cd = K.cd_sweep_rate(ctx, red, cw, r0["lam"] * n_total, sweeps=args.cd_sweeps)

    # ---- the whole leapfrog run the steps above are part of (KmerLrEstimator.Estimate, --lambda-auto=N from a cold
    # model: every epoch = one step + column sub-select + reduced solve), best of 3; reported beside the metric
    def estimate_leg(mat, lam_auto, reps):
        from kmerlr_amd import _driver
        best = None
        for _ in range(reps):
            est = _driver.hip_estimator(ctx, save_trace=1)
            R.fence()
            t0 = time.perf_counter()
            est.estimate(mat._h, lam_auto)
            te = R.max_over_ranks(time.perf_counter() - t0)
            lam_e, _nz = est.epochs()
            cl = est.classifiers()
            rec = {"seconds": round(te, 5), "epochs": len(lam_e), "cd_iterations": len(est.trace()),
                   "nonzero": int(np.count_nonzero(cl[-1]["theta"][1:])), "lambda": cl[-1]["lam"]}
            est.close()
            if best is None or rec["seconds"] < best["seconds"]:
                best = rec
        return best
    estimate = estimate_leg(full, [N], 3)
    in_situ = sweep_kernel_in_estimate(R, full, [N])

    ms_per_step = dt / args.steps * 1e3
    value = float(n_total) * float(m) / (dt / args.steps)          # samples x features / s, whole job

    lay = full.layout()
    alg_bytes = alg_bytes_of(lay, n_local, nnz, m)
    traffic, traffic_source = load_traffic(lay, nnz, m)
    roofline = roofline_of(lay, alg_bytes, kern_ms / max(launches, 1), launches, cold_ms, n_local, nnz, m, traffic, traffic_source)
    roofline["in_estimate"] = dict(in_situ, achieved=round(alg_bytes / (in_situ["kernel_avg_ms"] * 1e-3) / 1e9, 1) if in_situ["kernel_avg_ms"] > 0 else 0.0,
                                   frac=round(alg_bytes / (in_situ["kernel_avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if in_situ["kernel_avg_ms"] > 0 else 0.0)
    info = ctx.comm_info()
    ar_ms = ctx.allreduce_time(m + 6, dtype=1, iters=20) if R.comm else 0.0
    stats_c2 = ctx.comm_stats()
    probe_c2 = ctx.comm_probe()                      # what the context measured for THIS shape (a later shape measures again)

    configs3 = None
    if not args.no_configs3:                   # at EVERY N: the 1 -> 8 curve of the 1M-column shape needs all its points
        full.close(); red.close()
        full = red = None
        configs3 = configs3_leg(R, estimate_leg)

    cpu_baseline = None
    if rank == 0 and not args.no_cpu_baseline:
        cpu_baseline = cpu_baseline_leg(args, theta, cols, cw)

    if rank == 0:
        line = {
            "metric": "samples_x_features_per_s (full-width coordinate sweep: X.theta/sigmoid/loss + per-feature gradient + select)",
            "value": value, "unit": "samples*features/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "dtype_note": "row pass in f64; the full-width sweep accumulates in exact int64 fixed point", "data": "synthetic",
            "config": {"workload": "synthetic %d seqs/GPU x %d 8-mer columns, %d draws/row, --lambda-auto=%d "
                                   "(BASELINE configs[2])" % (n_local, m, draws, N),
                       "rows_per_gpu": n_local, "rows_total": n_total, "features": m, "nnz_per_gpu": nnz,
                       "selected": int(len(cols) - 1), "parallelism": "row-shards x%d, 1 collective of m+6 int64 per sweep" % world,
                       "transport": ("shm: all ranks on ONE card, host-staged collectives -- a rehearsal of the N > 1 code path, not a scaling figure"
                                     if R.shm else "rccl" if R.comm else "none")},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "comm": {"n_ranks_seen": (info["nranks"] if R.shm else info["rccl_ranks"]) if R.comm else 1, "allreduce_ms": round(ar_ms, 4),
                     "allreduce_elements": m + 6 if R.comm else 0, "sweep_collective": sweep_collective(world, m, N, probe_c2),
                     "selfcheck": "passed" if R.comm else "no communicator", "stats": stats_c2, "probe": probe_c2},
            "cd_sweeps_per_s": cd["sweeps_per_s"], "cd_sweeps": cd,
            "cd_solve": {"sweeps_to_converge": n_sweeps_solve, "seconds": round(t_solve, 4), "nonzero": int(np.count_nonzero(theta[1:]))},
            "estimate_lambda_auto": estimate,
            "step_check": {"lambda": out["lam"], "loss": out["loss"], "c": out["c"]},
            "setup_s": {"generate": round(t_gen, 2), "ingest_upload": round(t_up, 2)},
        }
        if configs3 is not None:
            line["configs3"] = configs3
        print(json.dumps(line), flush=True)
    if R.comm:
        R.dist.barrier()
    if full is not None:
        full.close(); red.close()
    R.close()


def sweep_kernel_in_estimate(R, mat, lam_auto):
    """The sweep kernel's own HIP-event time INSIDE a whole leapfrog run (every epoch: sweep, column sub-select, reduced
    solve with its IRLS / Gram / CD kernels in between): what the production epoch leaves in the Infinity Cache for the
    next sweep decides whether the resident or the cold figure applies."""
    from kmerlr_amd import _driver
    ctx = R.ctx
    est = _driver.hip_estimator(ctx)
    ctx.prof_enable(True)
    ctx.prof_read(reset=True)
    est.estimate(mat._h, lam_auto)
    launches, ms = ctx.prof_read(reset=True)
    ctx.prof_read_cd(reset=True)
    ctx.prof_enable(False)
    est.close()
    return {"kernel_avg_ms": round(ms / max(launches, 1), 4), "launches": launches,
            "note": "sweep kernel timed inside klr_estimator_estimate (sub-select and reduced solve between sweeps)"}


def sweep_collective(world, m, N, probe=None):
    """What klr_sweep_select posts per sweep (klr_select.hip): one rank nothing; from 32,768 columns on (KLR_RS_MIN_DIM)
    whichever the context MEASURED to be faster on this communicator (klr_ctx_comm_probe): a reduce-scatter of the int64
    column sums + an all-gather of nranks x 2N candidates, or one all-reduce; below that one all-reduce."""
    if world <= 1:
        return "none (one rank)"
    if m + 1 >= int(os.environ.get("KLR_RS_MIN_DIM", "32768")) and 2 * N <= 4096:
        if probe and probe.get("form") == "all-reduce" and probe.get("elements") == m + 6:
            return "all-reduce of %d int64 (measured faster than reduce-scatter + all-gather here)" % (m + 6)
        return "reduce-scatter of %d int64 + all-gather of %d x %d candidates" % (m + 6, world, 2 * N)
    return "all-reduce of %d int64" % (m + 6)


def configs3_leg(R, estimate_leg, n_local=125000, m=1048576, draws=993, N=500, warmup=3, steps=10):
    """BASELINE configs[3]: synthetic 1M x 1M over 8 GPUs = 125,000 x 1,048,576 per GPU (weak: rows grow with the
    number of ranks here), --lambda-auto=500.  Same step, same timing rules; the all-reduce is timed apart."""
    K, ctx, world, rank = R.K, R.ctx, R.world, R.rank
    n_total = n_local * world
    rowptr, col, val, labels = K.synth_csr(SEED, rank * n_local, n_local, n_total, m, draws, 2 * N)
    full = K.Matrix.from_csr(ctx, n_local, m + 1, rowptr, col, val, labels, n_total)
    del rowptr, col, val
    nnz = full.nnz
    cw = full.class_weights()
    r0 = K.sweep_select(ctx, full, None, None, np.zeros(1), cw, N)
    cols = np.nonzero(r0["b"])[0]
    red = full.select_columns(cols)
    theta, recs = K.cd_solve(ctx, red, np.zeros(len(cols)), cw=cw, l1reg=r0["lam"] * n_total, eps_loss=1e-8)
    stepper = K.SweepStep(ctx, full, red, cols, theta, cw, N, lambda_prev=r0["lam"] * n_total)
    dt, launches, kern_ms = R.timed_steps(stepper.run, warmup, steps)
    out = stepper.result()
    cold_ms = R.cold_kernel_ms(stepper.run, steps)
    lay = full.layout()
    alg_bytes = alg_bytes_of(lay, n_local, nnz, m)
    traffic, traffic_source = load_traffic(lay, nnz, m)
    roofline = roofline_of(lay, alg_bytes, kern_ms / max(launches, 1), launches, cold_ms, n_local, nnz, m, traffic, traffic_source)
    ar_ms = ctx.allreduce_time(m + 6, dtype=1, iters=10) if R.comm else 0.0
    est = estimate_leg(full, [N], 1)
    in_situ = sweep_kernel_in_estimate(R, full, [N])
    roofline["in_estimate"] = dict(in_situ, frac=round(alg_bytes / (in_situ["kernel_avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if in_situ["kernel_avg_ms"] > 0 else 0.0)
    info = ctx.comm_info()
    stats = ctx.comm_stats()
    full.close(); red.close()
    return {"config": {"workload": "synthetic %d seqs/GPU x %d k-mer columns, %d draws/row, --lambda-auto=%d (BASELINE configs[3]: "
                                   "1M x 1M over 8 GPUs)" % (n_local, m, draws, N),
                       "rows_per_gpu": n_local, "rows_total": n_total, "features": m, "nnz_per_gpu": nnz, "selected": int(len(cols) - 1)},
            "value": float(n_total) * float(m) / (dt / steps), "unit": "samples*features/s", "n_gpus": world,
            "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3, "scaling": "weak",
            "roofline": roofline,
            "comm": {"n_ranks_seen": (info["nranks"] if R.shm else info["rccl_ranks"]) if R.comm else 1, "allreduce_ms": round(ar_ms, 4),
                     "allreduce_elements": m + 6 if R.comm else 0,
                     "sweep_collective": sweep_collective(world, m, N, ctx.comm_probe()), "stats": stats, "probe": ctx.comm_probe()},
            "estimate_lambda_auto": est,
            "step_check": {"lambda": out["lam"], "loss": out["loss"], "c": out["c"]}}


def cpu_share():
    """Host cores this process may really use: the cgroup CPU quota when there is one (a GPU box hands a job a share of
    its cores), else the affinity mask."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(round(q / p))))
        except Exception:
            pass
    return n


def cpu_baseline_leg(args, theta, cols, cw):
    """Times the CPU restatement of the reference path (oracle/, plain C, reference loop order and 16 B/nnz row
    layout) on a bounded sample of the same workload: the first rows of the same synthetic matrix, same theta; one
    step = Gradient (a4) + Select (a6/a7).  `value` is the faithful single-thread figure (the reference's loops are
    single-threaded, kmerLr_logistic_regression.go:166,258); `all_cores` is the same loop over row blocks on every
    host core (OpenMP, per-thread gradients summed) -- what a multi-threaded port of the reference would reach."""
    from oracle import oracle as O
    nr = args.cpu_baseline_rows
    m, draws, N = args.features, args.draws, args.lambda_auto
    x, labels = O.synth_csr(SEED, 0, nr, nr, m, draws, 2 * N)
    th = np.zeros(m + 1)
    th[cols] = theta
    ncores = cpu_share()

    def timed(threads, budget):
        reps, t_tot = 0, 0.0
        while t_tot < budget and reps < 1000:
            t0 = time.perf_counter()
            g = O.gradient(x, labels, th, cw=cw, threads=threads)
            b = np.zeros(m + 1, dtype=np.uint8); b[0] = 1
            O.select(g[1:], b, 0, N)
            t_tot += time.perf_counter() - t0
            reps += 1
        return t_tot / reps, reps
    per1, reps1 = timed(1, 10.0)
    # the fastest thread count up to the share (a share smaller than the visible cores makes more threads slower)
    perN, repsN, threads = None, 0, 1
    for t in sorted({min(ncores, v) for v in (4, 8, 16, 32, 64, ncores)}):
        if t < 2 or nr // t < 256:
            continue
        pt, rt = timed(t, 1.5)
        if perN is None or pt < perN:
            perN, repsN, threads = pt, rt, t
    if perN is None:
        perN, repsN, threads = per1, reps1, 1
    return {"value": float(nr) * float(m) / per1, "unit": "samples*features/s", "cores": 1, "kind": "port",
            "sample": "first %d rows of the same synthetic matrix (%d nnz), %d repetitions of Gradient+Select, "
                      "CPU restatement of the Go loops (oracle/klr_oracle.c), 1 thread" % (nr, x.nnz, reps1),
            "ms_per_step_sample": per1 * 1e3,
            "all_cores": {"value": float(nr) * float(m) / perN, "cores": threads, "nproc": ncores, "repetitions": repsN,
                          "ms_per_step_sample": perN * 1e3,
                          "note": "same loops, rows split over all host cores with OpenMP (per-thread g, summed)"}}


if __name__ == "__main__":
    main()
