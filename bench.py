#!/usr/bin/env python
"""bench.py -- MINCO trajectory optimisations per second on the hill map (BASELINE.json metric).

step      = one pass of the hot path over one batch: uph_batch_solve, i.e. B full ALMTrajOpt::optimizeSE2Traj solves
            (initScaling + all ALM passes + all L-BFGS iterations) by one kernel launch, inputs resident in HBM.
workload  = hill scene (synthetic hill cloud; map built by the device plane-fit kernel before the timed
            region, x-slab sharded + all-gathered over RCCL when N > 1), B start/goal problems per GPU drawn with the
            config-3 protocol (seeds 1000 + rank*B + i, 3-10 m apart), parameters of run_hill.yaml.  Weak scaling: B per GPU fixed
            (default 16384: the tail of a launch -- workgroups finishing below full residency -- costs 18 % at 8192 and 8 % at 16384;
            the line also carries B = 256 / 4096 / 8192).  --workload km2: BASELINE.json configs[4] (analytic 1 km^2 terrain, fp32 cells).
value     = (B * n_gpus * K) / max-over-ranks wall time of K steps.
roofline  = dominant kernel uph_solver_kernel: algorithmic bytes of one launch / its HIP-event duration, against 8 TB/s HBM.
cpu_baseline = the CPU oracle (single thread, kind "port") on a bounded sample of the same batch, rank 0, N = 1 only.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_SAMPLE_EVAL = 47 * 8      # 24 s gather + 14 s duals/scales + 7 s residuals + ~2 s coefficients, s = 8 (fp64)
HBM_PEAK_GBS = 8000.0               # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def pmc_traffic(batch, stream_bytes):
    """HBM-side bytes per solve step from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json, written by tools/profile.sh;
    separate --pmc passes for FETCH_SIZE and WRITE_SIZE, both in KiB).  Calibration on known-byte microkernels in THIS kernel's
    access patterns (tools/micro/fetch_calib.hip, profiles/r02a_fetch_calibration.txt): contiguous reads -- 16-byte/lane rows and
    8-byte/lane streams alike -- are counted at exactly 1/2, an 8-byte random gather at 64 B per access (= the line it pulls),
    writes at 1.  The solve kernel's contiguous reads are the L-BFGS history rows and the per-sample duals / scales
    (`stream_bytes`, algorithmic): their uncounted half is added back; gathers and writes are taken as counted.
    None when no PMC summary for this batch size is committed."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            t = json.load(f)
        if int(t["batch"]) != int(batch):
            return None
        fetch = t["fetch_kib"] * 1024.0 / max(1, t["launches"])
        write = t["write_kib"] * 1024.0 / max(1, t["launches"])
        return fetch + min(fetch, 0.5 * stream_bytes) + write
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=0, help="trajectories per GPU per step (hill, default 16384) / in total (km2, default 4096)")
    ap.add_argument("--cpu-sample", type=int, default=256, help="problems solved by the CPU oracle for cpu_baseline (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the penalty-kernel and small-batch measurements (profiling runs)")
    ap.add_argument("--workload", choices=("hill", "km2"), default="hill",
                    help="hill: the BASELINE metric's scene (default).  km2: configs[4] -- analytic 1 km^2 fractal terrain in fp32 cells, --batch (default 4096) "
                         "local-goal solves in total, split over the ranks (strong scaling)")
    ap.add_argument("--map-size", type=float, default=1000.0, help="km2 workload: side of the square map [m]")
    ap.add_argument("--tiled", action="store_true", help="km2 workload, N > 1: every rank holds only its x-slab of the grid plus a 20 m halo and solves the problems that "
                                                         "start in its slab (owner routing, SURVEY.md 8e row 3) instead of replicating the grid")
    ap.add_argument("--fp32", action="store_true", help="fp32 arithmetic in the sample phase of the objective (uph_ctx_set_sample_precision(32); configs[4] \"fp32\"); "
                                                        "the line then says dtype \"f32 samples / f64 solver\"")
    ap.add_argument("--pipelined", action="store_true", help="additionally measure two contexts driven alternately with uph_batch_solve_async / uph_batch_wait "
                                                             "(the tail of one launch overlaps the head of the next); reported under \"pipelined\", the headline stays synchronous")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per trajectory (0 = automatic: 128 for large batches)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="also time the CPU oracle with this many threads (one trajectory per thread; context only)")
    args = ap.parse_args()
    km2 = args.workload == "km2"
    if not args.batch:
        args.batch = 4096 if km2 else 16384

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    distributed = world > 1 or os.environ.get("UPH_FORCE_DIST") == "1"      # the env knob exercises the RCCL path with one rank
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    device = local_rank if distributed else 0
    torch.cuda.set_device(device)

    import uneven_planner_amd as U
    from uneven_planner_amd import scenes

    # ---- map on the device (not timed).  N > 1: x-slabs + one RCCL all-gather
    gather = lambda full, slab: dist.all_gather_into_tensor(full, slab)
    t0 = time.time()
    if km2:
        # configs[4]: analytic fractal terrain, fp32 cells (16 bytes per cell; 1 km^2 at 0.25 m x 64 yaw bins = 16.4 GB, replicated per GPU,
        # or -- with --tiled -- one x-slab plus halo per GPU)
        from uneven_planner_amd.uneven_map import km2_map
        m = km2_map(args.map_size, rank, world, device, tiled=args.tiled, all_gather=gather if distributed else None)
    else:
        xyz = scenes.make_hill_cloud()      # hill cloud -> SE(2) grid by the plane-fit kernel
        m = U.UnevenMap(device=device)
        if distributed and int(m.voxel_num[0]) % world == 0:
            m.build_sharded(xyz, rank, world, gather)
        else:
            m.build(xyz)
    map_build_s = time.time() - t0
    map_stats = m.build_stats()

    # ---- problems, free cells only: config-3 protocol on the hill map; local goals (4..14 m) over the whole square for km2
    nx, ny = int(m.voxel_num[0]), int(m.voxel_num[1])
    gridinfo = (nx, ny, m.xy_resolution, m.map_origin[0], m.map_origin[1])
    if km2:
        total_batch = args.batch
        lo, args.batch = scenes.batch_share(total_batch, rank, world)      # this rank's share of the one batch (strong scaling)
        if args.batch < 1:
            raise SystemExit("km2: batch of %d cannot be split over %d ranks" % (total_batch, world))
        from uneven_planner_amd.uneven_map import km2_problems
        probs = km2_problems(m, args.map_size, args.batch, lo, rank, world)
    else:
        probs = scenes.random_problems(args.batch, seed0=1000 + rank * args.batch, occ_r2=m.occ_r2_buffer, grid=gridinfo)
    opt = U.ALMTrajOpt(m)
    if args.lanes:
        opt.set_lanes(args.lanes)
    if args.fp32:
        opt.set_sample_precision(32)
    opt.upload(probs)          # inputs resident in HBM from here on

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        opt.set_rho(1.0)
        opt.solve()
    # single-trajectory latency on the hill problem of configs[0]/[1] (not part of the timed region)
    single = U.ALMTrajOpt(m)
    single.upload([probs[0] if km2 else scenes.hill_problem()])
    single.set_rho(1.0); single.solve()
    single.set_rho(1.0); single.solve()
    sst = single.stats()
    single_ms = sst["kernel_ms"] + sst["prepare_ms"]
    single_ms_per_iter = sst["kernel_ms"] / max(1, sst["lbfgs_iters"])
    del single
    extras = {}
    if rank == 0 and not args.no_extras and not km2:
        # BASELINE configs[1]: the penalty kernel alone (uph_eval_batch: `repeat` objective+gradient evaluations per trajectory inside
        # one launch), on the hill trajectory x 256 and on the whole batch; algorithmic bytes = samples x 376 B
        R = 20
        pk = {}
        for tag, pp in (("hill_x256", [scenes.hill_problem()] * 256), ("batch", probs)):
            ev = U.ALMTrajOpt(m)
            if args.lanes:
                ev.set_lanes(args.lanes)
            ev.upload(pp)
            ev.init_scaling_batch()
            ev.eval_batch(None, repeat=R)
            ev.eval_batch(None, repeat=R)
            ms = ev.stats()["kernel_ms"]
            S = sum(s_["S"] for s_ in ev._sizes)
            gbs = S * R * BYTES_PER_SAMPLE_EVAL / (ms * 1e-3) / 1e9
            pk[tag] = {"trajectories": len(pp), "evals_per_launch": R, "kernel_ms": ms, "us_per_traj_eval": ms * 1e3 / R,
                       "M_traj_evals_per_s": len(pp) * R / ms / 1e3, "samples_per_traj": S / len(pp), "achieved_GBs": gbs, "frac": gbs / HBM_PEAK_GBS}
            del ev
        extras["penalty_kernel"] = pk
        # the batch sizes BASELINE.json names (configs[2]: 256, configs[4]: 4096) and the earlier rounds' 8192, same scene and protocol, one warm-up + three solves each
        for Bx in (256, 4096, 8192):
            if Bx >= args.batch:
                continue
            o2 = U.ALMTrajOpt(m)
            o2.upload(probs[:Bx])
            o2.set_rho(1.0); o2.solve()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                o2.set_rho(1.0); o2.solve()
            torch.cuda.synchronize()
            extras["traj_opts_per_s_B%d" % Bx] = Bx * 3 / (time.perf_counter() - t1)
            del o2
        # configs[4] "fp32" on the same scene: fp32 arithmetic in the sample phase (uph_ctx_set_sample_precision), B = 8192, three solves
        if args.batch >= 8192 and not args.fp32:
            o3 = U.ALMTrajOpt(m)
            o3.set_sample_precision(32)
            o3.upload(probs[:8192])
            o3.set_rho(1.0); o3.solve()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                o3.set_rho(1.0); o3.solve()
            torch.cuda.synchronize()
            extras["traj_opts_per_s_B8192_fp32_samples"] = 8192 * 3 / (time.perf_counter() - t1)
            del o3
    opt.upload(probs)

    kernel_ms, prepare_ms, evals, sample_evals, iters, hist_bytes = [], [], 0, 0, 0, 0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        opt.set_rho(1.0)       # every step does identical work (rho would otherwise persist across solves, Q7)
        opt.solve()
        st = opt.stats()
        kernel_ms.append(st["kernel_ms"]); prepare_ms.append(st["prepare_ms"])
        evals += st["evals"]; sample_evals += st["sample_evals"]; iters += st["lbfgs_iters"]; hist_bytes += st["hist_bytes"]
    barrier()
    dt = time.perf_counter() - t0
    if distributed:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    out = opt.download()
    rets = np.array([o["ret"] for o in out])
    pipelined = None
    if args.pipelined:
        # two contexts, each with its own stream and its own copy of the batch, launched alternately: K solves in total
        ctxs = [opt, U.ALMTrajOpt(m)]
        if args.lanes:
            ctxs[1].set_lanes(args.lanes)
        ctxs[1].upload(probs)
        for c_ in ctxs:
            c_.set_rho(1.0); c_.solve()
        barrier()
        t1 = time.perf_counter()
        ctxs[0].set_rho(1.0); ctxs[0].solve_async()
        for k_ in range(1, args.steps):
            ctxs[k_ % 2].set_rho(1.0); ctxs[k_ % 2].solve_async()
            ctxs[(k_ - 1) % 2].wait()
        ctxs[(args.steps - 1) % 2].wait()
        barrier()
        pdt = time.perf_counter() - t1
        if distributed:
            tm2 = torch.tensor([pdt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tm2, op=dist.ReduceOp.MAX)
            pdt = float(tm2.item())
        pipelined = {"value": args.batch * world * args.steps / pdt, "unit": "traj-opts/s", "contexts": 2, "ms_per_step": pdt / args.steps * 1e3}

    if rank == 0:
        K = args.steps
        total = (total_batch if km2 else args.batch * world) * K
        value = total / dt
        # km2: the gather reads fp32 cells (24 x 4 B per sample); duals, scales, residuals and coefficients stay fp64 (23 x 8 B)
        bytes_per_sample = (24 * 4 + 23 * 8) if km2 else BYTES_PER_SAMPLE_EVAL
        # roofline of the dominant kernel (per launch, rank 0): algorithmic bytes / HIP-event duration
        n_sum = sum(s["n"] for s in opt._sizes)
        per_launch_bytes = (sample_evals * bytes_per_sample + hist_bytes + iters * 2 * 8 * (n_sum / max(1, args.batch))) / K
        avg_ms = float(np.mean(kernel_ms))
        achieved = per_launch_bytes / (avg_ms * 1e-3) / 1e9
        res = {
            "metric": "MINCO traj-opts/sec (batch)", "value": value, "unit": "traj-opts/s", "n_gpus": world, "steps": K,
            "warmup": args.warmup, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "strong" if km2 else "weak",
            "vs_baseline": None, "dtype": "f32 samples / f64 solver" if args.fp32 else "f64", "data": "synthetic",
            "config": {"workload": ("configs[4]: analytic fractal terrain %.0f m x %.0f m (fBm H 0.8, seed 7), fp32 cell storage with fp64 arithmetic, one batch of %d "
                                    "local-goal (4-14 m) full ALM solves split over the GPUs, run_hill.yaml params" % (args.map_size, args.map_size, total_batch)) if km2 else
                                   ("hill scene (synthetic hill cloud, map built on device), batch of %d random start/goal "
                                    "full ALM solves per GPU (configs[1] scene, configs[2] start/goal protocol), run_hill.yaml params" % args.batch),
                       "batch_per_gpu": args.batch, "grid": [nx, ny, int(m.voxel_num[2])], "parallelism": ("dp%d" % world) + (" (grid tiled by x-slab owner + 20 m halo)" if km2 and m.tile is not None else "")},
            "ms_per_lbfgs_iter": single_ms_per_iter,       # single hill trajectory alone on the GPU (configs[1]): solve kernel ms / its L-BFGS iterations
            "single_traj_ms": single_ms, "batch_lbfgs_iters_per_s": iters / dt,
            "lbfgs_iters_per_traj": iters / K / args.batch, "evals_per_traj": evals / K / args.batch,
            "scaling_kernel_ms": float(np.mean(prepare_ms)),
            "converged_frac": float((rets == 0).mean()), "map_build_s": map_build_s, "map_kernel_ms": map_stats["kernel_ms"],
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None if km2 else pmc_traffic(args.batch, hist_bytes / K + sample_evals * 14 * 8 / K), "kernel": "uph_solver_kernel<%s,2> (ALM/L-BFGS solve)" % ("128,2" if args.batch >= 2304 else ("256,2" if args.batch >= 512 else "256,1")), "avg_launch_ms": avg_ms,
                         "algorithmic_bytes_per_launch": per_launch_bytes,
                         "sample_bytes_per_launch": sample_evals * bytes_per_sample / K, "history_bytes_per_launch": hist_bytes / K},
        }
        res.update(extras)
        if pipelined:
            res["pipelined"] = pipelined
        if world == 1 and not args.no_cpu and args.cpu_sample > 0:
            from oracle import oracle_py as O
            nsamp = min(args.cpu_sample, len(probs))
            if not km2:
                og = O.OracleGrid()
                og.set_cells(m.map_buffer)
            cdt, c_iters = 0.0, 0
            for p in probs[:nsamp]:
                # km2: the 1e9-cell grid stays on the device; the problem is solved on the window of cells around it, translated by whole
                # cells (window download not timed)
                g_, q_ = O.window_oracle(m, p)[:2] if km2 else (og, p)
                t0 = time.perf_counter()
                r = O.OracleALM(g_).optimize(q_)
                cdt += time.perf_counter() - t0
                c_iters += r["lbfgs_iters"]
            res["cpu_baseline"] = {"value": nsamp / cdt, "unit": "traj-opts/s", "cores": 1, "kind": "port",
                                   "sample": "first %d problems of the same batch, CPU oracle (C++ -O3, single thread), %.1f s" % (nsamp, cdt),
                                   "ms_per_lbfgs_iter": cdt * 1e3 / max(1, c_iters), "host_cpus": os.cpu_count()}
            if args.cpu_threads > 1 and not km2:
                # context only: the reference is single-threaded; this is "one trajectory per host thread" on the same box
                from concurrent.futures import ThreadPoolExecutor
                nmt = min(len(probs), max(nsamp, 8 * args.cpu_threads))
                t0 = time.perf_counter()
                with ThreadPoolExecutor(max_workers=args.cpu_threads) as ex:
                    list(ex.map(lambda p_: O.OracleALM(og).optimize(p_)["ret"], probs[:nmt]))
                mdt = time.perf_counter() - t0
                res["cpu_baseline_all_threads"] = {"value": nmt / mdt, "unit": "traj-opts/s", "cores": args.cpu_threads, "kind": "port",
                                                   "sample": "first %d problems, one trajectory per thread, %.1f s" % (nmt, mdt)}
    if distributed:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL leaves its version banner in the C stdio buffer, which would otherwise be flushed at exit, after the line below
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
