#!/usr/bin/env python3
"""bench.py -- full-width coordinate sweep of kmerLr's L1-LR hot path on MI355X.

One "step" = one leapfrog epoch's full-width pass (klr_sweep_select): X.theta / sigmoid / log-loss
over the rows (from the reduced matrix of the current theta), the per-feature gradient sweep
g = X'(cw (sigma - y))/n over the full CSC k-mer matrix, ONE all-reduce of [g | loss] across ranks,
and the top-2N |g| selection + lambda rule.  Inputs are resident in HBM before the timed region.

Workload (BASELINE.json configs[2]): synthetic 100k sequences x 65,536 8-mers per GPU (weak
scaling: rows grow with N), --lambda-auto=100, theta on 100 selected columns.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 20261004
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)


def parse():
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
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import kmerlr_amd as K

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d"
                             % (args.gpus, args.gpus))
        args.gpus = world
    torch.cuda.set_device(local_rank)
    comm = world > 1 or args.force_comm
    if comm:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    n_local, m, draws, N = args.rows_per_gpu, args.features, args.draws, args.lambda_auto
    n_total = n_local * world
    S = 2 * N

    ctx = K.Context(local_rank)
    if comm:
        # RCCL communicator owned by the library: one all-reduce per sweep on the library's stream
        ids = [K.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        ctx.init_rccl(ids[0], rank, world)

    # ---- synthetic shard, generated on the host and moved to HBM once -------------------------------
    t0 = time.time()
    rowptr, col, val, labels = K.synth_csr(SEED, rank * n_local, n_local, n_total, m, draws, S)
    t_gen = time.time() - t0
    t0 = time.time()
    full = K.Matrix.from_csr(ctx, n_local, m + 1, rowptr, col, val, labels, n_total)
    ctx.synchronize()
    t_up = time.time() - t0
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
    step = stepper.run

    for _ in range(args.warmup):
        step()
    ctx.synchronize()
    torch.cuda.synchronize()
    if comm:
        dist.barrier()
        torch.cuda.synchronize()
    ctx.prof_enable(True)
    ctx.prof_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    out = stepper.result()
    ctx.synchronize()
    torch.cuda.synchronize()
    if comm:
        dist.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    launches, kern_ms = ctx.prof_read(reset=True)
    ctx.prof_enable(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- reduced-problem CD sweep rate (sweeps/s): incremental cost of 64 more sweeps (soft threshold + loss
    # pass + stop checks each) inside one re-weighting, i.e. without the IRLS/Gram set-up
    def timed_solve(iters):
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            _, rr = K.cd_solve(ctx, red, np.zeros(len(cols)), cw=cw, l1reg=r0["lam"] * n_total, eps=0.0, eps_loss=0.0,
                               eval_loss=True, max_iter=iters)
            best = min(best, time.perf_counter() - t0)
        return best, len(rr)
    t_a, n_a = timed_solve(8)
    t_b, n_b = timed_solve(8 + args.cd_sweeps)
    cd_sweeps_per_s = (n_b - n_a) / (t_b - t_a) if t_b > t_a and n_b > n_a else 0.0

    # ---- the whole leapfrog run the steps above are part of (KmerLrEstimator.Estimate, --lambda-auto=N from a cold
    # model: every epoch = one step + column sub-select + reduced solve), best of 3; reported beside the metric
    estimate = None
    if world == 1:
        from kmerlr_amd import _driver
        best = None
        for _ in range(3):
            est = _driver.hip_estimator(ctx, save_trace=1)
            t0 = time.perf_counter()
            est.estimate(full._h, [N])
            te = time.perf_counter() - t0
            lam_e, _nz = est.epochs()
            cl = est.classifiers()
            rec = {"seconds": round(te, 5), "epochs": len(lam_e), "cd_iterations": len(est.trace()),
                   "nonzero": int(np.count_nonzero(cl[0]["theta"][1:])), "lambda": cl[0]["lam"]}
            est.close()
            if best is None or rec["seconds"] < best["seconds"]:
                best = rec
        estimate = best

    ms_per_step = dt / args.steps * 1e3
    value = float(n_total) * float(m) / (dt / args.steps)          # samples x features / s, whole job

    # ---- roofline of the dominant kernel (column sweep k_xt_s), per launch ----------------------------
    lay = full.layout()
    if lay["tiled"]:
        # single pass over the 2-D tiled packed stream: 2 B per unit entry (8-byte quads row:12 | 4 x col:12, a row's
        # run padded to 4 columns), 3 B (col:12|row:12) or 4 B (count:8|col:12|row:12) per entry, whichever the
        # ingest chose (value_bytes -1 / 0 / 1), padding included; one 32 KiB vector of int64 column sums written per
        # (tile, workgroup) piece,
        # s read once (the per-piece s tiles are L2 hits and not counted)
        alg_bytes = (3.0 + lay["value_bytes"]) * lay["padded_entries"] + lay["nseg"] * 4096.0 * 8 + 8.0 * n_local
        kname = "k_sweep_coo"
    else:
        # plain CSC: 4 B row id + 4|8 B value per stored nonzero, chunk table, s once per row, g once per column
        alg_bytes = (4.0 + lay["value_bytes"]) * nnz + 12.0 * lay["nseg"] + 8.0 * n_local + 8.0 * (m + 1)
        kname = "k_xt_s<%d>" % lay["lanes_per_segment"]
    # HBM traffic of the dominant kernel from the committed PMC passes (profiles/): valid for this exact workload
    traffic = None
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
        if (lay["tiled"] and pj["workload"]["nnz"] == nnz and pj["workload"]["features"] == m
                and pj["workload"].get("entry_bytes") == 3 + lay["value_bytes"]):
            traffic = pj["kernels"]["k_sweep_coo"]["hbm_traffic_bytes_per_launch"]
    except Exception:
        traffic = None
    kern_avg_ms = kern_ms / max(launches, 1)
    achieved = alg_bytes / (kern_avg_ms * 1e-3) / 1e9 if kern_avg_ms > 0 else 0.0
    two_pass_bytes = 16.0 * nnz + 8.0 * (n_local + 1) + 8.0 * (m + 1) + 9.0 * n_local + 8.0 * (m + 1)
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_source": "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE x2 for 16 B/lane loads per MI355X_MICROARCH.md; Infinity-Cache hits are counted)" if traffic else None,
                "kernel": kname, "layout": lay, "kernel_avg_ms": round(kern_avg_ms, 4), "launches": launches,
                "algorithmic_bytes_per_launch": alg_bytes,
                "survey_two_pass_bytes": two_pass_bytes,
                "survey_two_pass_equiv_gbs": round(two_pass_bytes / (kern_avg_ms * 1e-3) / 1e9, 1) if kern_avg_ms > 0 else 0.0}

    cpu_baseline = None
    if rank == 0 and not args.no_cpu_baseline and world == 1:
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
                       "selected": int(len(cols) - 1), "parallelism": "row-shards x%d, 1 RCCL all-reduce of m+6 int64 per sweep" % world},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "cd_sweeps_per_s": round(cd_sweeps_per_s, 1),
            "cd_solve": {"sweeps_to_converge": n_sweeps_solve, "seconds": round(t_solve, 4), "nonzero": int(np.count_nonzero(theta[1:]))},
            "estimate_lambda_auto": estimate,
            "step_check": {"lambda": out["lam"], "loss": out["loss"], "c": out["c"]},
            "setup_s": {"generate": round(t_gen, 2), "ingest_upload": round(t_up, 2)},
        }
        print(json.dumps(line), flush=True)
    if comm:
        dist.barrier()
    full.close(); red.close()
    ctx.close()
    if comm:
        dist.destroy_process_group()


def cpu_baseline_leg(args, theta, cols, cw):
    """Times the CPU restatement of the reference path (oracle/, plain C, 1 core, reference loop
    order and 16 B/nnz row layout) on a bounded sample of the same workload: the first rows of the
    same synthetic matrix, same theta; one step = Gradient (a4) + Select (a6/a7)."""
    from oracle import oracle as O
    nr = args.cpu_baseline_rows
    m, draws, N = args.features, args.draws, args.lambda_auto
    x, labels = O.synth_csr(SEED, 0, nr, nr, m, draws, 2 * N)
    th = np.zeros(m + 1)
    th[cols] = theta
    reps, t_tot = 0, 0.0
    while t_tot < 12.0 and reps < 1000:
        t0 = time.perf_counter()
        g = O.gradient(x, labels, th, cw=cw)
        b = np.zeros(m + 1, dtype=np.uint8); b[0] = 1
        O.select(g[1:], b, 0, N)
        t_tot += time.perf_counter() - t0
        reps += 1
    per = t_tot / reps
    return {"value": float(nr) * float(m) / per, "unit": "samples*features/s", "cores": 1, "kind": "port",
            "sample": "first %d rows of the same synthetic matrix (%d nnz), %d repetitions of Gradient+Select, "
                      "CPU restatement of the Go loops (oracle/klr_oracle.c), 1 thread" % (nr, x.nnz, reps),
            "ms_per_step_sample": per * 1e3}


if __name__ == "__main__":
    main()
