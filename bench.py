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


KERNEL_SOURCES = ("unevenhip.hip", "solver_program.hpp", "terrain_dev.hpp", "uph_common.hpp", "minco_op_host.hpp")


def kernel_sources_sha():
    """sha1 over the sources of the solve kernel: what a committed counter pass must have been taken from to describe this build"""
    import hashlib
    h = hashlib.sha1()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "uneven_planner_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


PMC_FILE = "pmc_traffic.json"      # (--workload astar: pmc_traffic_astar.json -- its own committed counter pass)


def pmc_counters(batch):
    """the committed counter pass (profiles/pmc_traffic*.json, written by tools/profile.sh) if it describes THIS kernel, workload and batch size, else (None, why)"""
    if LIVE_PMC is not None and int(LIVE_PMC.get("batch", -1)) == int(batch):
        return LIVE_PMC, None
    try:
        with open(os.path.join(ROOT, "profiles", PMC_FILE)) as f:
            t = json.load(f)
    except Exception:
        return None, "no profiles/" + PMC_FILE
    if int(t.get("batch", -1)) != int(batch):
        return None, "the committed counter pass (%s) is for B = %s, not for this batch size" % (t.get("tag"), t.get("batch"))
    if t.get("kernel_src_sha") != kernel_sources_sha():
        return None, "STALE: the committed counter pass (%s, kernel sources %s) is not of this build's kernel sources (%s) -- re-run tools/profile.sh" % (
            t.get("tag"), t.get("kernel_src_sha"), kernel_sources_sha())
    return t, None


LIVE_PMC = None      # filled by live_counter_passes(): the same dict shape as profiles/pmc_traffic*.json, measured by THIS invocation


def live_counter_passes(args, workload_flag):
    """rocprofv3 --pmc passes of this very command, run as children of this bench.py invocation (N = 1, default run): one pass per counter group, each a
    `bench.py --steps 1 --warmup 0 --no-cpu --no-extras --no-configs --no-live-counters` under `rocprofv3 --pmc ... ` (counters only: no tracing domains beside them),
    the solve kernel's rows (MODE 2) summed.  Returns a dict like profiles/pmc_traffic.json or None (no rocprofv3, a pass failed or timed out: the line then falls back to
    the committed pass).  ~20 s per pass."""
    import csv
    import re
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "no rocprofv3 on this box"
    groups = [("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU"),
              ("SQ_INSTS_VALU_MFMA_F64", "SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_VALU_MFMA_BUSY_CYCLES")]
    vals, launch_ms, t_all = {}, None, time.perf_counter()
    tmp = tempfile.mkdtemp(prefix="uph_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        for g in groups:
            out = os.path.join(tmp, g[0])
            cmd = [exe, "--pmc"] + list(g) + ["--output-format", "csv", "-d", out, "-o", "c", "--", sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "0",
                                               "--no-cpu", "--no-extras", "--no-configs", "--no-live-counters", "--batch", str(args.batch)] + workload_flag + (["--lanes", str(args.lanes)] if args.lanes else []) + (["--fp32"] if args.fp32 else [])
            try:
                r = subprocess.run(cmd, cwd=os.environ.get("TMPDIR", "/tmp"), capture_output=True, text=True, timeout=float(os.environ.get("UPH_BENCH_PMC_LIMIT", "150")))
            except subprocess.TimeoutExpired:
                return None, "the %s counter pass exceeded its limit" % g[0]
            if r.returncode != 0:
                return None, "the %s counter pass failed (rc %d): %s" % (g[0], r.returncode, (r.stderr or "")[-300:])
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if line and launch_ms is None:
                launch_ms = json.loads(line[-1])["roofline"]["avg_launch_ms"]      # (under the profiler: the counters' own launch)
            got = False
            for root_, _, files in os.walk(out):
                for f in files:
                    if f.endswith("counter_collection.csv"):
                        with open(os.path.join(root_, f)) as fh:
                            for row in csv.DictReader(fh):
                                k = row["Kernel_Name"].replace(" ", "")
                                if re.search(r"uph_solver_kernel<\d+,\d+,2(,(false|true))?>", k) or re.search(r"uph_solver_kernelILi\d+ELi\d+ELi2E", k):
                                    vals[row["Counter_Name"]] = vals.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                                    got = True
            if not got:
                return None, "the %s counter pass produced no rows of the solve kernel" % g[0]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return {"batch": args.batch, "tag": "rocprofv3 --pmc passes run by this bench.py invocation (%d passes, %.0f s)" % (len(groups), time.perf_counter() - t_all),
            "fetch_kib": vals.get("FETCH_SIZE", 0.0), "write_kib": vals.get("WRITE_SIZE", 0.0), "launches": 1, "sq_active_inst_valu": vals.get("SQ_ACTIVE_INST_VALU", 0.0),
            "sq_wave_cycles": vals.get("SQ_WAVE_CYCLES", 0.0), "sq_wait_any": vals.get("SQ_WAIT_ANY", 0.0), "sq_insts_valu": vals.get("SQ_INSTS_VALU", 0.0),
            "sq_insts_valu_mfma_f64": vals.get("SQ_INSTS_VALU_MFMA_F64", 0.0), "sq_insts_valu_mfma_mops_f64": vals.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0),
            "sq_valu_mfma_busy_cycles": vals.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), "launch_ms": launch_ms, "kernel_src_sha": kernel_sources_sha(), "git_head": "this run", "live": True}, None


def pmc_traffic(batch, stream_bytes):
    """HBM-side bytes per solve step from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json, written by tools/profile.sh;
    separate --pmc passes for FETCH_SIZE and WRITE_SIZE, both in KiB).  Calibration on known-byte microkernels in THIS kernel's
    access patterns (tools/micro/fetch_calib.hip, profiles/r02a_fetch_calibration.txt): contiguous reads -- 16-byte/lane rows and
    8-byte/lane streams alike -- are counted at exactly 1/2, an 8-byte random gather at 64 B per access (= the line it pulls),
    writes at 1.  The solve kernel's contiguous reads are the L-BFGS history rows and the per-sample duals / scales
    (`stream_bytes`, algorithmic): their uncounted half is added back; gathers and writes are taken as counted.
    None when no PMC summary for this batch size is committed."""
    t, why = pmc_counters(batch)
    if t is None:
        return None
    fetch = t["fetch_kib"] * 1024.0 / max(1, t["launches"])
    write = t["write_kib"] * 1024.0 / max(1, t["launches"])
    return fetch + min(fetch, 0.5 * stream_bytes) + write, t.get("tag", "profiles/pmc_traffic.json")


class Stage:
    """hard wall-clock limit for one stage of a run (a hung collective or a device that never answers must not park the job until the launcher's own
    limit): a watchdog thread ends THIS process with exit code 3 and a line on stderr naming the stage.  `with Stage("map build", 300): ...`"""
    def __init__(self, name, seconds, rank=0):
        self.name, self.seconds, self.rank = name, float(os.environ.get("UPH_BENCH_STAGE_LIMIT", seconds)), rank
        self.t0 = 0.0

    def __enter__(self):
        import threading
        self.t0 = time.perf_counter()
        self.done = threading.Event()

        def watch():
            if not self.done.wait(self.seconds):
                sys.stderr.write("bench.py: rank %d: stage \"%s\" exceeded its limit of %.0f s -- giving up (exit 3)\n" % (self.rank, self.name, self.seconds))
                sys.stderr.flush()
                os._exit(3)
        threading.Thread(target=watch, daemon=True).start()
        return self

    def __exit__(self, *exc):
        self.done.set()
        self.elapsed = time.perf_counter() - self.t0
        return False


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` outside a launcher: start N ranks of this script (one process per GPU, RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in their environment, the same contract torch.distributed.run provides), relay rank 0's JSON line, fail if any rank fails.  A rank that
    dies takes the others with it (they would otherwise wait for it at the next barrier until the collective's own timeout), and the whole job has a
    wall-clock limit (UPH_BENCH_JOB_LIMIT seconds, default 1500)."""
    import socket
    import subprocess
    import tempfile
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), UPH_BENCH_SPAWNED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=None, text=True))
    limit = float(os.environ.get("UPH_BENCH_JOB_LIMIT", "1500"))
    t0 = time.time()
    why = None
    while True:
        codes = [q.poll() for q in procs]
        if all(c is not None for c in codes):
            break
        bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if bad or time.time() - t0 > limit:
            why = ("rank(s) %s exited with %s" % (bad, [codes[r] for r in bad])) if bad else ("the job exceeded %.0f s" % limit)
            for q in procs:              # (exactly the processes started above)
                if q.poll() is None:
                    q.kill()
            codes = [q.wait() for q in procs]
            break
        time.sleep(0.2)
    out0.seek(0)
    text = out0.read()
    line = [ln for ln in text.splitlines() if ln.startswith("{")]
    if why or any(codes) or not line:
        sys.stderr.write("bench.py: %s; ranks exited with %s\n%s\n" % (why or "a rank failed", codes, text))
        raise SystemExit(1)
    print(line[-1], flush=True)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def single_process(args):
    """--single-process: what a one-process host (the reference's ROS node) gets from N GPUs through the C-ABI alone.  Map: uph_map_build_multi
    (x-slab fits on N host threads + one in-library ncclAllGather).  Batch: one context per device, every device its own B problems
    (weak scaling, seeds 1000 + g B + i as in the multi-process form), uph_batch_solve_async on all, then uph_batch_wait on all."""
    import torch
    import uneven_planner_amd as U
    from uneven_planner_amd import scenes
    N = args.gpus
    if torch.cuda.device_count() < N:
        raise SystemExit("bench.py --single-process: --gpus %d but only %d GPU(s) visible" % (N, torch.cuda.device_count()))
    B = args.batch or 16384
    import ctypes as C
    import hashlib
    # the in-library RCCL binding and an all-gather of a known pattern over the clique of the N devices, before anything depends on it
    selftest = None
    if N > 1:
        with Stage("uph_rccl_selftest", 240) as stg:
            info = C.create_string_buffer(256)
            rc = U._lib.load().uph_rccl_selftest(N, info, 256)
        selftest = {"ok": rc == 0, "world": N, "ms": stg.elapsed * 1e3, "info": info.value.decode(errors="replace")}
        if rc != 0:
            raise SystemExit("bench.py --single-process: uph_rccl_selftest(%d) failed: %s" % (N, U._lib.load().uph_last_error()))
    xyz = scenes.make_hill_cloud()
    maps = [U.UnevenMap(device=g) for g in range(N)]
    with Stage("uph_map_build_multi", 300):
        t0 = time.time()
        stages = U.UnevenMap.build_multi(maps, xyz)
        map_build_s = time.time() - t0
        t0 = time.time()
        stages2 = U.UnevenMap.build_multi(maps, xyz, download=False)      # second call: the RCCL clique is cached
        map_build_warm_s = time.time() - t0
    hashes = [hashlib.sha1(np.ascontiguousarray(mm.map_buffer).tobytes()).hexdigest()[:15] for mm in maps]
    if len(set(hashes)) != 1:
        raise SystemExit("bench.py --single-process: the devices hold different grids after uph_map_build_multi: %s" % hashes)
    if N > 1 and not stages2.get("via_rccl"):
        sys.stderr.write("bench.py --single-process: the slab exchange did NOT run as an RCCL collective (device-to-device copies instead)\n")
    m0 = maps[0]
    nx, ny = int(m0.voxel_num[0]), int(m0.voxel_num[1])
    gridinfo = (nx, ny, m0.xy_resolution, m0.map_origin[0], m0.map_origin[1])
    opts, sizes = [], []
    for g in range(N):
        pr = scenes.random_problems(B, seed0=1000 + g * B, occ_r2=m0.occ_r2_buffer, grid=gridinfo)
        o = U.ALMTrajOpt(maps[g])
        if args.lanes:
            o.set_lanes(args.lanes)
        o.upload(pr)
        opts.append(o)

    def sync():
        for g in range(N):
            torch.cuda.synchronize(g)

    def step():
        for o in opts:
            o.set_rho(1.0); o.solve_async()
        for o in opts:
            o.wait()
    with Stage("warm-up", 120 + 10 * args.warmup):
        for _ in range(args.warmup):
            step()
        sync()
    with Stage("timed region", 60 + 10 * args.steps):
        t0 = time.perf_counter()
        kms = []
        for _ in range(args.steps):
            step()
            kms.append([o.stats()["kernel_ms"] for o in opts])
        sync()
        dt = time.perf_counter() - t0
    rets = np.array([r["ret"] for r in opts[0].download(full=False)])
    res = single_process_line(args, N, B, dt, np.mean(np.array(kms), axis=0), float((rets == 0).mean()), [nx, ny, int(m0.voxel_num[2])], stages, stages2, map_build_s, map_build_warm_s,
                              [mm.build_stats()["stages_ms"] for mm in maps])
    res["rccl_selftest"] = selftest
    res["map_hash"] = hashes[0]
    res["map_hash_identical_on_all_devices"] = True
    kmean = np.mean(np.array(kms), axis=0)
    res["per_gpu_kernel_spread"] = float((kmean.max() - kmean.min()) / kmean.max())
    print(json.dumps(res), flush=True)


def single_process_line(args, N, B, dt, per_gpu_kernel_ms, converged_frac, grid, stages, stages2, map_build_s, map_build_warm_s, per_device_stages):
    """the JSON line of --single-process (one place, so that the CPU tier's dry run of an 8-GPU node prints the line's real shape)"""
    return {"metric": "MINCO traj-opts/sec (batch)", "value": B * N * args.steps / dt, "unit": "traj-opts/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "hill scene (synthetic hill cloud, map built on the devices by uph_map_build_multi), batch of %d random start/goal full ALM solves per GPU, run_hill.yaml params" % B,
                       "batch_per_gpu": B, "grid": grid, "launcher": "single process, C-ABI multi-GPU entries (no torch.distributed)",
                       "rccl_world": N if stages2.get("via_rccl") else 1, "parallelism": "dp%d" % N},
            "per_rank_ms_per_step": [dt / args.steps * 1e3] * N,
            "per_gpu_kernel_ms": [float(v) for v in per_gpu_kernel_ms], "converged_frac": converged_frac,
            "map_build_multi": dict(first_call_s=map_build_s, warm_call_s=map_build_warm_s, first=stages, warm=stages2,
                                    per_device_slab_stages_ms=per_device_stages,
                                    note="first call includes the RCCL clique creation (~1 s) and the scratch allocations; the timed region never contains a map build")}


def single_process_dry_run(args):
    """UPH_BENCH_SPAWN_ECHO=1 with --single-process (CPU tier, tests/test_dist_cpu.py): every host-side decision the N-GPU run takes -- the x-slab
    each device fits and whether the all-gather runs in place (uph_multi_slab_plan), the seeds of each device's problems -- and the line's shape,
    without touching a device"""
    import ctypes as C
    import uneven_planner_amd as U
    L = U._lib.load()
    N, B = args.gpus, args.batch or 16384
    nx = int(math.ceil(10.0 / 0.05))                                         # run_hill.yaml: map_size_x / xy_resolution (uneven_map.cpp:108)
    x0, x1, per, inpl = (C.c_int32 * N)(), (C.c_int32 * N)(), C.c_int32(0), C.c_int32(0)
    U._lib.check(L.uph_multi_slab_plan(nx, N, x0, x1, C.byref(per), C.byref(inpl)), "uph_multi_slab_plan")
    fake = {"fit_ms": 0.0, "exchange_ms": 0.0, "commit_ms": 0.0, "exchange_device_ms": 0.0, "via_rccl": N > 1}
    line = single_process_line(args, N, B, 1.0, [0.0] * N, 0.0, [nx, nx, 64], fake, fake, 0.0, 0.0, [None] * N)
    line["dry_run"] = {"slabs": [[int(x0[g]), int(x1[g])] for g in range(N)], "rows_per_slab": int(per.value), "all_gather_in_place": bool(inpl.value),
                       "seed0_per_device": [1000 + g * B for g in range(N)]}
    print(json.dumps(line), flush=True)


def astar_batch(U, scenes, m, gridinfo, want, seed0):
    """The batch the goal -> trajectory chain produces (the reference's rcvWpsCallBack, plan_manager.cpp:55-134): KinoAstar::plan on the device for random hill
    goals (config-3 protocol, seeds from seed0), PlanManager's resampling stage (uph_resample_batch) on every path found; exactly `want` problems (goals are
    drawn until that many searches have succeeded; ~4 % of the random goals have no path and are dropped as the reference drops them)."""
    from uneven_planner_amd import resample as RS
    ka = U.KinoAstar(m, slots=4096)
    probs, goals, found, t_search = [], 0, 0, 0.0
    while len(probs) < want:
        nq = max(256, int((want - len(probs)) / 0.94) + 64)
        S_, G_ = scenes.random_queries(nq, seed0=seed0 + goals, occ_r2=m.occ_r2_buffer, grid=gridinfo)
        sr = ka.plan_batch(S_, G_, path_cap=768)
        t_search += ka.stats()["kernel_ms"] * 1e-3
        paths = [q_["path"] for q_ in sr if q_["status"] == 0 and q_["n_path"] <= 768]
        goals += nq; found += len(paths)
        probs += RS.resample_batch(paths)
    del ka
    return probs[:want], {"goals_drawn": goals, "paths_found": found, "search_kernel_s": t_search}


def solve_rate(torch, opt, B, K, bytes_per_sample=BYTES_PER_SAMPLE_EVAL):
    """one warm-up + K timed uph_batch_solve steps of the uploaded batch: throughput, kernel time and the solve kernel's own roofline figure"""
    opt.set_rho(1.0); opt.solve()
    torch.cuda.synchronize()
    ms, prep, se, hb, it, ev = [], [], 0, 0, 0, 0
    t1 = time.perf_counter()
    for _ in range(K):
        opt.set_rho(1.0); opt.solve()
        st = opt.stats()
        ms.append(st["kernel_ms"]); prep.append(st["prepare_ms"]); se += st["sample_evals"]; hb += st["hist_bytes"]; it += st["lbfgs_iters"]; ev += st["evals"]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t1
    n_mean = sum(s_["n"] for s_ in opt._sizes) / max(1, B)
    by = (se * bytes_per_sample + hb + it * 2 * 8 * n_mean) / K
    ach = by / (float(np.mean(ms)) * 1e-3) / 1e9
    out = opt.download(full=False)
    return {"value": B * K / dt, "unit": "traj-opts/s", "batch": B, "steps": K, "ms_per_step": dt / K * 1e3, "scaling_kernel_ms": float(np.mean(prep)),
            "lbfgs_iters_per_traj": it / K / B, "evals_per_traj": ev / K / B, "converged_frac": float(np.mean([o_["ret"] == 0 for o_ in out])),
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                         "avg_launch_ms": float(np.mean(ms)), "algorithmic_bytes_per_launch": by, "bytes_per_sample_evaluation": bytes_per_sample}}, out


def first_eval_parity(U, O, m, og, probs, n=8, window=False):
    """first objective evaluation after initScaling -- f and the gradient at x0 -- of the first n problems: device (uph_init_scaling_batch + uph_eval_batch)
    against the CPU oracle (the checker: this leg belongs to cpu_baseline / parity, never to a timed region)"""
    pp = probs[:n]
    ev = U.ALMTrajOpt(m)
    ev.upload(pp)
    ev.init_scaling_batch()
    f, g = ev.eval_batch(ev.x0_packed(pp))
    ef, eg = 0.0, 0.0
    for i, p in enumerate(pp):
        g_, q_, _ = O.window_oracle(m, p) if window else (og, p, None)
        a = O.OracleALM(g_)
        x0 = a.setup(q_)
        a.init_scaling(x0)
        fo, go, _ = a.eval(x0)
        ef = max(ef, abs(f[i] - fo) / abs(fo))
        eg = max(eg, float(np.abs(g[i] - go).max() / np.abs(go).max()))
    return {"problems": len(pp), "f_rel_max": ef, "grad_rel_max": eg, "tolerance": 1e-9, "ok": bool(ef < 1e-9 and eg < 1e-9),
            "what": "f and grad f of innerCallback at x0 after initScaling, device vs CPU oracle" + (" on the window of cells around each problem" if window else "")}


def oracle_baseline(O, make_grid_and_problem, probs, budget_s, what):
    """the CPU oracle (single thread, kind "port") on the first problems of a batch until ~budget_s seconds are spent"""
    cdt, n, its, conv = 0.0, 0, 0, 0
    for p in probs:
        g_, q_ = make_grid_and_problem(p)
        t0 = time.perf_counter()
        r = O.OracleALM(g_).optimize(q_)
        cdt += time.perf_counter() - t0
        n += 1; its += r["lbfgs_iters"]; conv += int(r["ret"] == 0)
        if cdt > budget_s:
            break
    return {"value": n / cdt, "unit": "traj-opts/s", "cores": 1, "kind": "port", "sample": "first %d problems of %s, CPU oracle (C++ -O3, single thread), %.1f s" % (n, what, cdt),
            "ms_per_lbfgs_iter": cdt * 1e3 / max(1, its), "converged_frac": conv / n, "cpu_model": cpu_model()}


def measure_configs(args, torch, U, scenes, m_hill, single_line, pk):
    """Every BASELINE.json config in the driver-run line: one entry per config -- its workload, the measured value, the kernel's
    roofline figure, a CPU baseline on a bounded sample and a parity statement.  Rank 0, one GPU; none of it is inside the headline's timed region."""
    from oracle import oracle_py as O
    try:
        names = json.load(open(os.path.join(ROOT, "BASELINE.json")))["configs"]
    except Exception:
        names = ["configs[%d]" % i for i in range(5)]
    golden = os.path.join(ROOT, "tests", "golden")
    cfg = []
    t_all = time.perf_counter()
    # ---- [0] hill, single goal, the reference's CPU back-end = the CPU oracle on that goal (the plumbing case)
    og = O.OracleGrid()
    og.set_cells(m_hill.map_buffer)
    hp = scenes.hill_problem()
    ts, r0 = [], None
    for _ in range(3):
        t0 = time.perf_counter()
        r0 = O.OracleALM(og).optimize(hp)
        ts.append(time.perf_counter() - t0)
    a = O.OracleALM(og)
    x0 = a.setup(hp)
    a.init_scaling(x0)
    t0 = time.perf_counter()
    for _ in range(50):
        a.eval(x0)
    cpu_eval_us = (time.perf_counter() - t0) / 50 * 1e6
    cfg.append({"config": names[0], "value": min(ts) * 1e3, "unit": "ms per optimizeSE2Traj call", "higher_is_better": False, "device": "host CPU, one thread (the oracle = kind \"port\"; the reference needs Eigen / ROS and cannot be built here)",
                "lbfgs_iters": r0["lbfgs_iters"], "ret": r0["ret"], "ms_per_lbfgs_iter": min(ts) * 1e3 / max(1, r0["lbfgs_iters"]), "us_per_objective_evaluation": cpu_eval_us,
                "cpu_baseline": {"value": 1.0 / min(ts), "unit": "traj-opts/s", "cores": 1, "kind": "port", "sample": "the hill goal, best of 3 solves", "cpu_model": cpu_model()}})
    # ---- [1] hill, one GPU, single trajectory, batched penalty kernel
    c1 = {"config": names[1], "value": single_line["single_traj_ms"], "unit": "ms per optimizeSE2Traj call (one trajectory alone on the GPU, scaling + solve kernels)", "higher_is_better": False,
          "lbfgs_iters": single_line["single_traj_lbfgs_iters"], "ms_per_lbfgs_iter": single_line["ms_per_lbfgs_iter"],
          "cpu_baseline": {"value": min(ts) * 1e3, "unit": "ms per call", "cores": 1, "kind": "port", "sample": "the same goal, CPU oracle, best of 3", "us_per_objective_evaluation": cpu_eval_us}}
    if pk:
        h = pk["hill_x256"]
        c1["penalty_kernel"] = {"workload": "the hill trajectory x 256, 20 evaluations per launch", "us_per_objective_evaluation": h["kernel_ms"] * 1e3 / h["evals_per_launch"],
                                "roofline": {"bound": "hbm", "achieved": h["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": h["frac"], "frac_moved": h["frac_moved"], "frac_a5_only": h["a5_only"]["frac"], "traffic": None}}
    cfg.append(c1)
    # ---- [2] desert (the reference's own cloud), batch of 256 random start / goal solves
    try:
        xyz = np.load(os.path.join(golden, "desert_xyz.npz"))["xyz"]
        md = U.UnevenMap(device=m_hill.device)
        md.build(xyz)
        nx, ny = int(md.voxel_num[0]), int(md.voxel_num[1])
        pd = scenes.random_problems(256, seed0=1000, occ_r2=md.occ_r2_buffer, grid=(nx, ny, md.xy_resolution, md.map_origin[0], md.map_origin[1]))
        od = U.ALMTrajOpt(md)
        od.upload(pd)
        line, _ = solve_rate(torch, od, 256, 3)
        ogd = O.OracleGrid()
        ogd.set_cells(md.map_buffer)
        line.update({"config": names[2], "workload": "uneven_map/maps/desert.pcd (fixture tests/golden/desert_xyz.npz), map built on the device, seeds 1000..1255, 3-10 m apart, run_hill.yaml parameters, full ALM solves",
                     "map_build_ms": md.build_stats()["stages_ms"]["call"], "parity": first_eval_parity(U, O, md, ogd, pd, 8),
                     "cpu_baseline": oracle_baseline(O, lambda p_: (ogd, p_), pd, 6.0, "the same batch")})
        cfg.append(line)
        del od, md
    except Exception as e:
        cfg.append({"config": names[2], "error": repr(e)})
    # ---- [3] volcano: the SE(2) grid plane-fit build.  One GPU here; the 8-GPU x-slab exchange is played by 8 maps on this device (bit-identical grid required)
    try:
        xyz = np.load(os.path.join(golden, "vocano_xyz.npz"))["xyz"]
        prm = dict(max_rho=0.08)                      # run_vocano.yaml differs from run_hill.yaml in max_rho only
        mv = U.UnevenMap(prm, device=m_hill.device)
        mv.build(xyz, download=False)
        tb = []
        for _ in range(3):
            t0 = time.perf_counter()
            mv.build(xyz, download=False)
            tb.append(time.perf_counter() - t0)
        st = mv.build_stats()
        mv.download()
        cells = st["cell_iters"]
        # ~87 candidate points x 12 B + 32 B out per cell-iteration at this density
        by = cells * (87 * 12 + 32)
        world = 8
        maps = [U.UnevenMap(prm, device=m_hill.device) for _ in range(world)]
        U.UnevenMap.build_multi(maps, xyz, download=False)
        t0 = time.perf_counter()
        stages = U.UnevenMap.build_multi(maps, xyz, download=False)
        multi_s = time.perf_counter() - t0
        same = True
        for k_ in (0, world - 1):
            maps[k_].download()
            same = same and bool(np.array_equal(maps[k_].map_buffer, mv.map_buffer) and np.array_equal(maps[k_].occ_buffer, mv.occ_buffer))
        del maps
        # CPU baseline: the oracle's constructMap on two x-rows of the same cloud, projected to the 200 rows of the grid
        ob = O.OracleMapBuilder(xyz=xyz)
        ogv = O.OracleGrid()
        rows = 2
        t0 = time.perf_counter()
        ob.construct(ogv, map_params=dict(mv.params), x0=100, x1=100 + rows, do_occ=False)
        crow = time.perf_counter() - t0
        co, _ = ogv.get_cells()
        sl = slice(100 * int(mv.voxel_num[1]) * int(mv.voxel_num[2]), (100 + rows) * int(mv.voxel_num[1]) * int(mv.voxel_num[2]))
        dcell = np.abs(mv.map_buffer.reshape(-1, 4)[sl] - co[sl]).max(axis=1)
        cfg.append({"config": names[3], "value": min(tb) * 1e3, "unit": "ms per uph_map_build call (200 x 200 x 64 cells x 2 iterations, one GPU: upload, crop + voxel, bucketing, plane fits, commit)", "higher_is_better": False,
                    "workload": "uneven_map/maps/vocano.pcd (fixture tests/golden/vocano_xyz.npz), run_vocano.yaml", "kernel_ms": st["kernel_ms"], "stages_ms": st["stages_ms"], "cell_iterations": cells,
                    "M_cell_iterations_per_s": cells / (st["kernel_ms"] * 1e-3) / 1e6,
                    "roofline": {"bound": "hbm", "achieved": by / (st["kernel_ms"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by / (st["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                 "algorithmic_bytes_per_launch": by, "note": "algorithmic neighbourhood bytes (SURVEY.md 8d: 1.08 KB per cell-iteration); the kernel stages each (x, y) column's disc once in LDS and is bound by the arithmetic of the fits"},
                    "world_of_8_on_one_device": {"slabs": world, "call_ms": multi_s * 1e3, "fit_ms": stages["fit_ms"], "exchange_ms": stages["exchange_ms"], "commit_ms": stages["commit_ms"], "via_rccl": stages["via_rccl"],
                                                 "grid_bit_identical_to_the_single_build": same,
                                                 "note": "uph_map_build_multi with 8 maps on this one device: the x-slab rule, the slab exchange (device-to-device copies -- RCCL refuses one device twice) and the commit of an 8-GPU build; the 8-device run itself is the driver's"},
                    "parity": {"cells_compared": int(dcell.size), "max_abs_diff_vs_oracle": float(dcell.max()), "frac_within_1e-9": float((dcell < 1e-9).mean()), "ok": bool((dcell < 1e-9).mean() >= 0.999)},
                    "cpu_baseline": {"value": crow / rows * int(mv.voxel_num[0]) * 1e3, "unit": "ms per build (projected)", "cores": 1, "kind": "port", "sample": "the oracle's constructMap on x-rows 100..%d of the same cloud (%.2f s), projected to %d rows" % (100 + rows - 1, crow, int(mv.voxel_num[0]))}})
        if not same:
            raise SystemExit("bench.py: the world-of-8 map build differs from the single build")
        del mv
    except SystemExit:
        raise
    except Exception as e:
        cfg.append({"config": names[3], "error": repr(e)})
    # ---- [4] 1 km^2 analytic fractal terrain in fp32 cells, batch of 4096 local goals: fp64 arithmetic and fp32 sample arithmetic
    try:
        from uneven_planner_amd.uneven_map import km2_map, km2_problems
        t0 = time.perf_counter()
        mk = km2_map(1000.0, 0, 1, m_hill.device)
        fill_s = time.perf_counter() - t0
        pk_ = km2_problems(mk, 1000.0, 4096, 0, 0, 1)
        bps = 24 * 4 + 23 * 8                      # the gather reads fp32 cells; duals, scales, residuals, coefficients stay fp64
        ok_ = U.ALMTrajOpt(mk)
        ok_.upload(pk_)
        l64, o64 = solve_rate(torch, ok_, 4096, 3, bps)
        del ok_
        o32 = U.ALMTrajOpt(mk)
        o32.set_sample_precision(32)
        o32.upload(pk_)
        l32, _ = solve_rate(torch, o32, 4096, 3, bps)
        del o32
        nsamp = 12
        cb = oracle_baseline(O, lambda p_: O.window_oracle(mk, p_)[:2], pk_[:nsamp], 8.0, "the same batch, each on the window of cells around it")
        rel = []
        for p_, d_ in zip(pk_[:nsamp], o64[:nsamp]):
            g_, q_, sh = O.window_oracle(mk, p_)
            r_ = O.OracleALM(g_).optimize(q_)
            xo = np.array(r_["x"], dtype=np.float64).copy()
            nin = np.asarray(p_["inner_xy"]).reshape(2, -1).shape[1]
            xo[1:1 + 2 * nin:2] += sh[0]; xo[2:2 + 2 * nin:2] += sh[1]
            rel.append((float(np.abs(d_["x"] - xo).max() / max(1e-300, np.abs(xo).max())), r_["lbfgs_iters"], int(d_["ret"] == r_["ret"])))
        l64.update({"config": names[4], "workload": "analytic fBm terrain 1000 m x 1000 m at 0.25 m x 64 yaw bins = 1.02e9 fp32 cells (16.4 GB, filled on the device in %.1f s), 4096 local goals 4-14 m apart, "
                                                    "each solved in its own local frame, fp64 arithmetic; one GPU here (the batch splits over the ranks for N > 1: bench.py --workload km2 --gpus N)" % fill_s,
                    "dtype": "fp32 cell storage, f64 arithmetic", "map_fill_s": fill_s,
                    "fp32_sample_arithmetic": {k_: l32[k_] for k_ in ("value", "unit", "ms_per_step", "converged_frac", "roofline")},
                    "parity": dict(first_eval_parity(U, O, mk, None, pk_, 6, window=True), final_waypoints_le_1e4=float(np.mean([r_[0] <= 1e-4 for r_ in rel])), same_ret=float(np.mean([r_[2] for r_ in rel])),
                                   final_sample=len(rel), note="final way-points of full solves against the window oracle: the optimiser is chaotic, the fraction decays with the oracle's iteration count (DESIGN.md 6)"),
                    "cpu_baseline": cb})
        cfg.append(l64)
        del mk
    except Exception as e:
        cfg.append({"config": names[4], "error": repr(e)})
    return {"seconds": time.perf_counter() - t_all, "entries": cfg}


def dist_selftest(torch, dist, rank, world, tdev):
    """collective self-test on the run's own communicator, before anything is measured: an all-gather of the rank ids and a sum with known answers"""
    with Stage("collective self-test", 120, rank) as stg:
        mine = torch.full((1024,), float(rank), dtype=torch.float64, device=tdev)
        allr = torch.empty(1024 * world, dtype=torch.float64, device=tdev)
        dist.all_gather_into_tensor(allr, mine)
        tot = mine.clone()
        dist.all_reduce(tot)
        if tdev != "cpu":
            torch.cuda.synchronize()
        ok = bool(torch.equal(allr.view(world, 1024)[:, 0].cpu(), torch.arange(world, dtype=torch.float64))) and float(tot[0].item()) == world * (world - 1) / 2.0
    if not ok:
        raise SystemExit("bench.py: rank %d: the collective self-test returned wrong data" % rank)
    return {"ok": ok, "world": dist.get_world_size(), "ms": stg.elapsed * 1e3, "what": "all_gather_into_tensor of the rank ids + all_reduce(sum) on the run's own communicator"}


def exchange_map_hash(torch, dist, rank, world, tdev, map_hash):
    """every rank's hash of its gathered grid, all-gathered and compared: a rank that holds another grid ends the job"""
    with Stage("map hash exchange", 120, rank):
        hv = torch.tensor([int(map_hash, 16)], dtype=torch.int64, device=tdev)
        hall = torch.empty(world, dtype=torch.int64, device=tdev)
        dist.all_gather_into_tensor(hall, hv)
        if tdev != "cpu":
            torch.cuda.synchronize()
    if not bool((hall == hall[0]).all().item()):
        raise SystemExit("bench.py: rank %d: the gathered map differs between ranks (%s)" % (rank, [hex(int(v)) for v in hall.tolist()]))
    return True


def gather_times(torch, dist, world, tdev, dt, steps):
    """per-rank step times (reporting) and the MAX over the ranks (the job's time)"""
    mine = torch.tensor([dt], dtype=torch.float64, device=tdev)
    allt = torch.empty(world, dtype=torch.float64, device=tdev)
    dist.all_gather_into_tensor(allt, mine)
    tmax = torch.tensor([dt], dtype=torch.float64, device=tdev)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    return [float(v) / steps * 1e3 for v in allt.tolist()], float(tmax.item())


def dry_distributed(args):
    """UPH_BENCH_DRY_DIST=1 under a launcher (CPU tier, tests/test_dist_cpu.py): the rank handshake of an N-GPU run -- world == --gpus, the collective self-test, the
    map-hash exchange, the barrier-bracketed timing reduction -- over gloo on CPU tensors, no device touched: the Python of the N > 1 path runs for worlds 2 .. 8 before
    it ever meets eight GPUs.  UPH_BENCH_DRY_BAD_RANK=r makes rank r hold another grid (the job must fail)."""
    import datetime
    import hashlib
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: the launcher's world and the flag disagree" % (args.gpus, world))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    with Stage("rendezvous", 120, rank):
        dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=120))
    st = dist_selftest(torch, dist, rank, world, "cpu")
    cells = np.arange(4096, dtype=np.float64) + (1.0 if os.environ.get("UPH_BENCH_DRY_BAD_RANK") == str(rank) else 0.0)
    h = hashlib.sha1(cells.tobytes()).hexdigest()[:15]
    same = exchange_map_hash(torch, dist, rank, world, "cpu", h)
    dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (1 + rank))
    dist.barrier()
    per_rank, dt = gather_times(torch, dist, world, "cpu", time.perf_counter() - t0, args.steps)
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "rccl_selftest": st, "map_hash": h, "map_hash_identical_on_all_ranks": same, "per_rank_ms_per_step": per_rank,
                          "per_rank_spread": (max(per_rank) - min(per_rank)) / max(per_rank), "ms_per_step": dt / args.steps * 1e3}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=0, help="trajectories per GPU per step (hill, default 16384) / in total (km2, default 4096)")
    ap.add_argument("--cpu-sample", type=int, default=256, help="problems solved by the CPU oracle for cpu_baseline (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the penalty-kernel and small-batch measurements (profiling runs)")
    ap.add_argument("--extras-multi", action="store_true", help="run rank 0's single-GPU extras (penalty kernel, small batches, front end ...) even when N > 1 (default: N = 1 only)")
    ap.add_argument("--no-live-counters", action="store_true", help="do not run the rocprofv3 --pmc passes of this command as children (N = 1 default run); roofline.traffic / valu_busy / mfma then "
                                                                     "come from the committed counter pass under profiles/, if it matches this build")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-config block (BASELINE.json configs[0..4]: desert B = 256, volcano map build, km^2 B = 4096 fp64 / fp32 ...)")
    ap.add_argument("--workload", choices=("hill", "astar", "km2"), default="hill",
                    help="hill: the BASELINE metric's scene, Hermite stand-in initial paths (default).  astar: the same scene and goal protocol with the initial paths the "
                         "reference's own chain produces -- KinoAstar::plan on the device + PlanManager's resampling -- instead of the stand-in (--batch solves per GPU, default 16384).  "
                         "km2: configs[4] -- analytic 1 km^2 fractal terrain in fp32 cells, --batch (default 4096) local-goal solves in total, split over the ranks (strong scaling)")
    ap.add_argument("--map-size", type=float, default=1000.0, help="km2 workload: side of the square map [m]")
    ap.add_argument("--tiled", action="store_true", help="km2 workload, N > 1: every rank holds only its x-slab of the grid plus a 20 m halo and solves the problems that "
                                                         "start in its slab (owner routing, SURVEY.md 8e row 3) instead of replicating the grid")
    ap.add_argument("--fp32", action="store_true", help="fp32 arithmetic in the sample phase of the objective (uph_ctx_set_sample_precision(32); configs[4] \"fp32\"); "
                                                        "the line then says dtype \"f32 samples / f64 solver\"")
    ap.add_argument("--pipelined", action="store_true", help="additionally measure two contexts driven alternately with uph_batch_solve_async / uph_batch_wait "
                                                             "(the tail of one launch overlaps the head of the next); reported under \"pipelined\", the headline stays synchronous")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per trajectory (0 = automatic: 128 for large batches)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="also time the CPU oracle with this many threads (one trajectory per thread; context only)")
    ap.add_argument("--single-process", action="store_true", help="N > 1 without a launcher and without torch.distributed: ONE process drives all --gpus devices through "
                                                                  "the C-ABI's own multi-GPU entries (uph_map_build_multi: host threads + in-library RCCL all-gather; one context per device "
                                                                  "with uph_batch_solve_async) -- what a single-process host such as the reference's ROS node would do")
    args = ap.parse_args()
    if os.environ.get("UPH_BENCH_SPAWN_ECHO") == "1" and "WORLD_SIZE" in os.environ:      # CPU-tier test of spawn_ranks (tests/test_dist_cpu.py): report the rank environment and leave
        print(json.dumps({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")} | {"gpus": args.gpus}), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.single_process:
        spawn_ranks(args.gpus, sys.argv[1:])
        return
    if os.environ.get("UPH_BENCH_DRY_DIST") == "1" and "WORLD_SIZE" in os.environ and not args.single_process:
        return dry_distributed(args)
    if args.single_process:
        if os.environ.get("UPH_BENCH_SPAWN_ECHO") == "1":
            return single_process_dry_run(args)
        return single_process(args)
    km2 = args.workload == "km2"
    astar_wl = args.workload == "astar"
    if astar_wl:
        global PMC_FILE
        PMC_FILE = "pmc_traffic_astar.json"
    map_device_s = map_device_warm_s = map_download_s = None
    if not args.batch:
        args.batch = 4096 if km2 else 16384

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # N copies of an N = 1 run (or an N-rank job reported as something else) must not pass for a scaling point
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: the launcher's world and the flag disagree" % (args.gpus, world))
    import torch
    import torch.distributed as dist
    if torch.cuda.device_count() < max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world))):
        raise SystemExit("bench.py: %d ranks on this node but only %d GPU(s) visible" % (world, torch.cuda.device_count()))
    distributed = world > 1 or os.environ.get("UPH_FORCE_DIST") == "1"      # the env knob exercises the RCCL path with one rank
    selftest = None
    if distributed:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        with Stage("rendezvous + RCCL communicator", 240, rank):
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=240))
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: RCCL world of %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))
        selftest = dist_selftest(torch, dist, rank, world, "cuda")
    device = local_rank if distributed else 0
    torch.cuda.set_device(device)

    import uneven_planner_amd as U
    from uneven_planner_amd import scenes

    # ---- map on the device (not timed).  N > 1: x-slabs + one RCCL all-gather
    gather = lambda full, slab: dist.all_gather_into_tensor(full, slab)
    stg_map = Stage("map build (+ slab all-gather)", 900 if km2 else 300, rank)
    stg_map.__enter__()
    t0 = time.time()
    if km2:
        # configs[4]: analytic fractal terrain, fp32 cells (16 bytes per cell; 1 km^2 at 0.25 m x 64 yaw bins = 16.4 GB, replicated per GPU,
        # or -- with --tiled -- one x-slab plus halo per GPU)
        from uneven_planner_amd.uneven_map import km2_map
        m = km2_map(args.map_size, rank, world, device, tiled=args.tiled, all_gather=gather if distributed else None)
    else:
        xyz = scenes.make_hill_cloud()      # hill cloud -> SE(2) grid by the plane-fit kernel
        m = U.UnevenMap(device=device)
        t0 = time.time()
        if distributed and int(m.voxel_num[0]) % world == 0:
            m.build_sharded(xyz, rank, world, gather)
        else:
            m.build(xyz, download=False)    # uph_map_build: cloud upload, crop + voxel filter, bucketing, plane fits, commit -- all on the device
            map_device_s = time.time() - t0
            m.build(xyz, download=False)    # (second call: the build scratch exists)
            map_device_warm_s = time.time() - t0 - map_device_s
            td = time.time()
            m.download()                    # host copies of map_buffer / c_buffer / occupancy for the untouched host consumers: 105 MB into fresh numpy arrays
            map_download_s = time.time() - td
    map_build_s = time.time() - t0
    stg_map.__exit__(None, None, None)
    map_stats = m.build_stats()
    # every rank must hold the SAME grid before anything is solved on it: a hash of the host copy of the gathered cells, compared over the ranks
    map_hash, map_hash_same = None, None
    if not km2:
        import hashlib
        map_hash = hashlib.sha1(np.ascontiguousarray(m.map_buffer).tobytes()).hexdigest()[:15]
        if distributed:
            map_hash_same = exchange_map_hash(torch, dist, rank, world, "cuda", map_hash)

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
    elif astar_wl:
        with Stage("front end: searches + resampling for the A*-seeded batch", 300, rank):
            probs, astar_meta = astar_batch(U, scenes, m, gridinfo, args.batch, 1000 + rank * 2 * args.batch)
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

    with Stage("warm-up (%d steps)" % args.warmup, 120 + 10 * args.warmup, rank):
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
    single_iters = sst["lbfgs_iters"]
    del single
    extras = {}
    if rank == 0 and not args.no_extras and not km2 and not astar_wl and (world == 1 or args.extras_multi):      # N > 1: the other ranks would sit at the barrier for the ~50 s these take
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
            # the same launch by the bytes it really moves: the 7 residual stores per sample of SURVEY.md 8d's definition are performed ONCE per launch here
            # (refreshResiduals after the R evaluations), not once per evaluation
            gbs_moved = (S * R * (BYTES_PER_SAMPLE_EVAL - 7 * 8) + S * 7 * 8) / (ms * 1e-3) / 1e9
            # A5 ALONE (uph_penalty_batch: calConstrainCostGrad, alm_traj_opt.cpp:663-991 -- resident coefficients -> cost, gdC, gdT and, like the reference's
            # function on every call, hx / gx): samples + scatter only, no MINCO generate / adjoint.  All 376 defined bytes are moved: defined = moved
            ev.penalty_batch(repeat=R, store_residuals=True)
            ev.penalty_batch(repeat=R, store_residuals=True)
            ms_a5 = ev.stats()["kernel_ms"]
            ev.penalty_batch(repeat=R, store_residuals=False)
            ms_a5n = ev.stats()["kernel_ms"]
            gbs_a5 = S * R * BYTES_PER_SAMPLE_EVAL / (ms_a5 * 1e-3) / 1e9
            pk[tag] = {"trajectories": len(pp), "evals_per_launch": R, "kernel_ms": ms, "us_per_traj_eval": ms * 1e3 / R,
                       "M_traj_evals_per_s": len(pp) * R / ms / 1e3, "samples_per_traj": S / len(pp), "achieved_GBs": gbs, "frac": gbs / HBM_PEAK_GBS,
                       "frac_moved": gbs_moved / HBM_PEAK_GBS,
                       "a5_only": {"kernel_ms": ms_a5, "achieved_GBs": gbs_a5, "frac": gbs_a5 / HBM_PEAK_GBS,
                                   "kernel_ms_without_residual_stores": ms_a5n,
                                   "frac_without_residual_stores_by_bytes_moved": S * R * (BYTES_PER_SAMPLE_EVAL - 7 * 8) / (ms_a5n * 1e-3) / 1e9 / HBM_PEAK_GBS}}
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
        # twice the headline's batch (same protocol, seeds continued): what the launch tail -- workgroups finishing below full residency, 7 % of a launch at 16384 -- is worth
        if args.batch == 16384:
            pb2 = probs + scenes.random_problems(args.batch, seed0=1000 + 8 * args.batch, occ_r2=m.occ_r2_buffer, grid=gridinfo)
            o4 = U.ALMTrajOpt(m)
            o4.upload(pb2)
            o4.set_rho(1.0); o4.solve()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(2):
                o4.set_rho(1.0); o4.solve()
            torch.cuda.synchronize()
            extras["traj_opts_per_s_B%d" % len(pb2)] = len(pb2) * 2 / (time.perf_counter() - t1)
            del o4, pb2
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
        # the boundary call itself: ONE uph_optimize_batch = upload + solve + download of x and the coefficients into pageable host arrays, the
        # reference's synchronous contract (alm_traj_opt.h:92-98, result pulled afterwards :165-168); B = the batch and B = 1
        bo = U.ALMTrajOpt(m)
        if args.lanes:
            bo.set_lanes(args.lanes)
        bd = {}
        for tag, pp in (("B%d" % args.batch, probs), ("B1", [scenes.hill_problem()])):
            prep = bo.prepare_boundary(pp)
            bo.set_rho(1.0); bo.optimize_boundary(pp, prepared=prep)
            ts = []
            for _ in range(2 if len(pp) > 1 else 5):
                bo.set_rho(1.0); bo.optimize_boundary(pp, prepared=prep)
                ts.append(bo.last_boundary_s)
            bd[tag] = {"traj_opts_per_s": len(pp) / min(ts), "ms_per_call": min(ts) * 1e3}
        bd["note"] = "uph_optimize_batch wall clock (upload + initScaling + solve + download of x, c_xy, c_yaw), pageable host arrays, best of the repeats"
        extras["boundary"] = bd
        del bo
        # the front end that produces the back-end's inputs.  KinoAstar::plan (kino_astar.cpp:67-236) for the start / goal pairs
        # of the batch's first problems (same streams, same acceptance), one wave64 per query (csrc/kino_search.hip); the CPU restatement is timed
        # in the cpu_baseline leg below
        try:
            if world > 1:
                raise RuntimeError("single-GPU measurement: skipped when the job spans several GPUs (run `python bench.py` for it)")
            fq = 65536 if args.batch >= 16384 else min(2048, args.batch)      # many queries per workspace: the shared cursor balances the launch's tail
            S_, G_ = scenes.random_queries(fq, seed0=1000, occ_r2=m.occ_r2_buffer, grid=gridinfo)
            ka = U.KinoAstar(m, slots=min(fq, 4096))      # explicit workspaces (one per wave slot of 256 CUs): the automatic ones would grow inside the timed calls
            ka.plan_batch(S_[:64], G_[:64], path_cap=1)
            fe = {"slots": ka.slots, "primitives": ka.n_primitives}
            for nq in sorted(set((1, 256, min(fq, 2048), fq))):
                t1 = time.perf_counter()
                # (the largest batch bounds its download: paths clipped at 64 poses, the searches themselves are complete -- status, counters, n_path; the goal ->
                # trajectory chain below pulls whole paths)
                r_ = ka.plan_batch(S_[:nq], G_[:nq], path_cap=64 if nq > 4096 else 512, complete=nq <= 4096)
                wall = time.perf_counter() - t1
                fe["B%d" % nq] = {"queries_per_s": nq / wall, "ms_per_call": wall * 1e3, "kernel_ms": ka.stats()["kernel_ms"], "found": float(np.mean([q_["status"] == 0 for q_ in r_])),
                                  "expansions_per_query": float(np.mean([q_["iter_num"] for q_ in r_])), "M_expansions_per_s": float(np.sum([q_["iter_num"] for q_ in r_])) / wall / 1e6}
            extras["front_end"] = fe
            extras["_front_end_queries"] = (S_, G_)
            # goal -> trajectory with every stage batched: search (device), PlanManager's resampling stage (uph_resample_batch, host C++), optimise
            # (device) -- the reference's rcvWpsCallBack chain (plan_manager.cpp:56-134) for `pb` goals in one go; goals whose search fails are dropped
            # as the reference drops them (empty front_end_path).  Wall clock of the three stages, results downloaded.
            from uneven_planner_amd import resample as RS
            pb = min(16384, fq)
            t1 = time.perf_counter()
            sr = ka.plan_batch(S_[:pb], G_[:pb], path_cap=768)
            t2 = time.perf_counter()
            search_kernel_s = ka.stats()["kernel_ms"] * 1e-3
            paths = [q_["path"] for q_ in sr if q_["status"] == 0 and q_["n_path"] <= 768]
            t2b = time.perf_counter()
            pr_ = RS.resample_batch(paths)
            t3 = time.perf_counter()
            po = U.ALMTrajOpt(m)
            po.set_rho(1.0)
            prep = po.prepare_boundary(pr_)                       # (ctypes packing of the Python binding: a C++ host hands its arrays over as they are)
            t3b = time.perf_counter()
            pout = po.optimize_boundary(pr_, prepared=prep)
            t4 = time.perf_counter()
            first_call_s = po.last_boundary_s
            po.set_rho(1.0)
            po.optimize_boundary(pr_, prepared=prep)              # the same call again: device buffers and MINCO operators of this context exist now
            native = search_kernel_s + (t3 - t2b) + po.last_boundary_s
            extras["pipeline"] = {"goals": pb, "paths_found": len(paths), "search_kernel_s": search_kernel_s, "resample_s": t3 - t2b, "optimise_call_s": po.last_boundary_s,
                                  "optimise_first_call_s": first_call_s,
                                  "goals_per_s": pb / native, "trajectories_per_s": len(paths) / native, "python_binding_overhead_s": (t4 - t1) - (search_kernel_s + (t3 - t2b) + first_call_s),
                                  "converged_frac": float(np.mean([o_["ret"] == 0 for o_ in pout])), "mean_pieces": float(np.mean([p_["inner_xy"].shape[1] + 1 for p_ in pr_])),
                                  "note": "KinoAstar::plan -> PlanManager resampling -> ALMTrajOpt::optimizeSE2Traj for a batch of goals: search kernel + uph_resample_batch (host C++) + "
                                          "one uph_optimize_batch call (upload, initScaling, solve, download; the second call of its context -- the first one, which also allocates the "
                                          "device buffers and builds the MINCO operators, is optimise_first_call_s); the ctypes packing / unpacking of the Python binding is reported separately"}
            del po
            del ka
            # the headline quoted on REAL front-end paths as well.  The batch = the problems the goal -> trajectory chain above produced
            # (KinoAstar::plan -> PlanManager's resampling, plan_manager.cpp:55-134) instead of the Hermite stand-in paths of the headline; same step
            # definition (uph_batch_solve, inputs resident in HBM, rho reset), its own roofline from its own counters; parity buckets in the cpu leg below
            ao = U.ALMTrajOpt(m)
            if args.lanes:
                ao.set_lanes(args.lanes)
            ao.upload(pr_)
            ao.set_rho(1.0); ao.solve()
            torch.cuda.synchronize()
            KA = 5
            a_ms, a_prep, a_se, a_hb, a_it, a_ev = [], [], 0, 0, 0, 0
            t1 = time.perf_counter()
            for _ in range(KA):
                ao.set_rho(1.0); ao.solve()
                st_ = ao.stats()
                a_ms.append(st_["kernel_ms"]); a_prep.append(st_["prepare_ms"]); a_se += st_["sample_evals"]; a_hb += st_["hist_bytes"]; a_it += st_["lbfgs_iters"]; a_ev += st_["evals"]
            torch.cuda.synchronize()
            a_dt = time.perf_counter() - t1
            a_out = ao.download(full=False)
            a_n = sum(s_["n"] for s_ in ao._sizes)
            a_bytes = (a_se * BYTES_PER_SAMPLE_EVAL + a_hb + a_it * 2 * 8 * (a_n / max(1, len(pr_)))) / KA
            a_ach = a_bytes / (float(np.mean(a_ms)) * 1e-3) / 1e9
            extras["astar_seeded"] = {"metric": "MINCO traj-opts/sec (batch), A*-seeded inputs", "value": len(pr_) * KA / a_dt, "unit": "traj-opts/s", "batch": len(pr_), "steps": KA,
                                      "ms_per_step": a_dt / KA * 1e3, "converged_frac": float(np.mean([o_["ret"] == 0 for o_ in a_out])),
                                      "lbfgs_iters_per_traj": a_it / KA / len(pr_), "evals_per_traj": a_ev / KA / len(pr_), "mean_pieces": float(np.mean([p_["inner_xy"].shape[1] + 1 for p_ in pr_])),
                                      "scaling_kernel_ms": float(np.mean(a_prep)),
                                      "roofline": {"bound": "hbm", "achieved": a_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a_ach / HBM_PEAK_GBS, "traffic": None,
                                                   "avg_launch_ms": float(np.mean(a_ms)), "algorithmic_bytes_per_launch": a_bytes},
                                      "workload": "the %d of %d random hill goals whose KinoAstar::plan search found a path, resampled by PlanManager's stage (piece_len 0.3 ...), full ALM solves" % (len(pr_), pb)}
            extras["traj_opts_per_s_astar_seeded"] = extras["astar_seeded"]["value"]
            extras["_astar"] = (pr_, a_out)
            del ao
        except Exception as e:
            extras["front_end"] = {"error": repr(e)}
    opt.upload(probs)

    kernel_ms, prepare_ms, evals, sample_evals, iters, hist_bytes = [], [], 0, 0, 0, 0
    with Stage("timed region (%d steps)" % args.steps, 60 + 10 * args.steps, rank):
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
    per_rank_ms = [dt / args.steps * 1e3]
    if distributed:
        per_rank_ms, dt = gather_times(torch, dist, world, "cuda", dt, args.steps)      # (reporting + the MAX over the ranks: outside the timed region)

    out = opt.download(full=False)
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
        live_why = None
        if world == 1 and not km2 and not args.no_live_counters and not args.no_extras and not args.no_cpu:
            # driver-observed counters: the --pmc passes of this command run here, now; a failure falls back to the committed pass
            global LIVE_PMC
            try:
                LIVE_PMC, live_why = live_counter_passes(args, ["--workload", "astar"] if astar_wl else [])
            except Exception as e:
                LIVE_PMC, live_why = None, repr(e)
        no_pass = "km2 workload: no committed counter pass" if km2 else ("fp32 sample arithmetic: the committed counter pass is of the fp64 kernels" if (args.fp32 and LIVE_PMC is None) else None)
        tr = None if no_pass else pmc_traffic(args.batch, hist_bytes / K + sample_evals * 14 * 8 / K)
        pc, pc_why = (None, no_pass) if no_pass else pmc_counters(args.batch)
        if tr and LIVE_PMC is not None:
            traffic, traffic_src = tr[0], "%s; FETCH_SIZE calibrated as profiles/*_fetch_calibration.txt prescribes (contiguous reads counted at 1/2)" % tr[1]
        else:
            traffic, traffic_src = (tr[0], "committed rocprofv3 --pmc passes of this command (%s), not measured in this run%s" % (tr[1], ("; live passes: " + live_why) if live_why else "")) if tr else (None, (pc_why or "") + (("; live passes: " + live_why) if live_why else ""))
        # the bound that actually binds: VALU issue.  SQ_ACTIVE_INST_VALU counts quad-cycles summed over all waves; per SIMD
        # (4 per CU, 256 CUs) and per launch of the committed pass, against that pass's own launch duration at the 2.4 GHz shader clock
        valu_busy = wait_frac = None
        mfma = None
        if pc is not None and pc.get("sq_active_inst_valu") and pc.get("launch_ms"):
            simd_cycles = 256 * 4 * pc["launch_ms"] * 1e-3 * 2.4e9
            valu_busy = pc["sq_active_inst_valu"] * 4.0 / max(1, pc["launches"]) / simd_cycles
            wait_frac = pc["sq_wait_any"] / pc["sq_wave_cycles"] if pc.get("sq_wave_cycles") else None
            if pc.get("sq_insts_valu_mfma_f64"):
                # north_star: "(where used) MFMA utilisation reported from rocprof against gfx950 peak".  The one matrix-core contraction of the path is the
                # xy half of the gradient scatter (v_mfma_f64_16x16x4_f64: 2048 flop, 64 cycles per instruction = 32 flop/clk/SIMD, tools/micro/mfma_f64_probe.hip)
                mf = pc["sq_insts_valu_mfma_f64"] / max(1, pc["launches"]) * 2048.0
                mfma = {"insts_per_launch": pc["sq_insts_valu_mfma_f64"] / max(1, pc["launches"]), "instruction": "v_mfma_f64_16x16x4_f64", "achieved": mf / (pc["launch_ms"] * 1e-3) / 1e12,
                        "peak": 32.0 * 1024 * 2.4e9 / 1e12, "unit": "TFLOP/s",
                        "util": mf / (pc["launch_ms"] * 1e-3) / (32.0 * 1024 * 2.4e9),
                        "busy_frac_of_simd_cycles": (pc.get("sq_valu_mfma_busy_cycles", 0.0) / max(1, pc["launches"])) / simd_cycles,
                        "note": "fp64 matrix-core peak = 32 flop/clk/SIMD x 1024 SIMDs x 2.4 GHz = 78.6 TFLOP/s (measured issue rate of the instruction); the path is not a GEMM -- the "
                                "matrix cores carry the one shared-operand dense contraction it contains (DESIGN.md 7b)"}
        # 7 of the 47 doubles per sample are the residual stores of SURVEY.md 8d's definition, which the solve kernel performs once per L-BFGS
        # pass, not per evaluation: the fraction without them is reported next to the defined one
        moved_bytes = (sample_evals * (bytes_per_sample - 7 * 8) + hist_bytes + iters * 2 * 8 * (n_sum / max(1, args.batch))) / K
        res = {
            "metric": "MINCO traj-opts/sec (batch)", "value": value, "unit": "traj-opts/s", "n_gpus": world, "steps": K,
            "warmup": args.warmup, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "strong" if km2 else "weak",
            "vs_baseline": None, "dtype": "f32 samples / f64 solver" if args.fp32 else "f64", "data": "synthetic",
            "config": {"workload": ("configs[4]: analytic fractal terrain %.0f m x %.0f m (fBm H 0.8, seed 7), fp32 cell storage with fp64 arithmetic, one batch of %d "
                                    "local-goal (4-14 m) full ALM solves split over the GPUs, run_hill.yaml params" % (args.map_size, args.map_size, total_batch)) if km2 else
                                   (("hill scene (synthetic hill cloud, map built on device), A*-seeded batch: %d problems per GPU = random start/goal pairs (configs[2] protocol) whose "
                                     "KinoAstar::plan search (device) found a path, resampled by PlanManager's stage (piece_len 0.3, yaw_piece_times 2, mean_vel 0.5 ...), "
                                     "full ALM solves, run_hill.yaml params" % args.batch) if astar_wl else
                                    ("hill scene (synthetic hill cloud, map built on device), batch of %d random start/goal "
                                     "full ALM solves per GPU (configs[1] scene, configs[2] start/goal protocol), run_hill.yaml params" % args.batch)),
                       "batch_per_gpu": args.batch, "grid": [nx, ny, int(m.voxel_num[2])], "launcher": "self-spawned ranks" if os.environ.get("UPH_BENCH_SPAWNED") == "1" else ("torch.distributed.run" if distributed else "single process"),
                       "rccl_world": dist.get_world_size() if distributed else 1, "parallelism": ("dp%d" % world) + (" (grid tiled by x-slab owner + 20 m halo)" if km2 and m.tile is not None else "")},
            "per_rank_ms_per_step": per_rank_ms, "per_rank_spread": (max(per_rank_ms) - min(per_rank_ms)) / max(per_rank_ms),
            "rccl_selftest": selftest, "map_hash": map_hash, "map_hash_identical_on_all_ranks": map_hash_same,
            "ms_per_lbfgs_iter": single_ms_per_iter,       # single hill trajectory alone on the GPU (configs[1]): solve kernel ms / its L-BFGS iterations
            "single_traj_ms": single_ms, "single_traj_lbfgs_iters": single_iters,      # (the solve is chaotic: a rounding-level change of the arithmetic moves this ONE problem's iteration count -- 216 / 253 in round 5 -- read ms_per_lbfgs_iter for the kernel)
            "batch_lbfgs_iters_per_s": iters / dt,
            "lbfgs_iters_per_traj": iters / K / args.batch, "evals_per_traj": evals / K / args.batch,
            "scaling_kernel_ms": float(np.mean(prepare_ms)),
            "converged_frac": float((rets == 0).mean()), "map_build_s": map_build_s, "map_kernel_ms": map_stats["kernel_ms"],
            "map_build_device_s": map_device_s, "map_build_device_warm_s": map_device_warm_s, "map_download_s": map_download_s,      # N = 1 hill: first / second uph_map_build call, then the D2H copy
            "map_build_stages_ms": map_stats.get("stages_ms"),      # rank 0's uph_map_build (its x-slab when N > 1): upload, crop + voxel, bucketing, kernel, commit, call; map_build_s adds the all-gather and the download into numpy
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "kernel": "uph_solver_kernel<%s,2> (ALM/L-BFGS solve)" % ("128,2" if args.batch >= 2304 else ("256,2" if args.batch >= 512 else "256,1")), "avg_launch_ms": avg_ms,
                         "algorithmic_bytes_per_launch": per_launch_bytes, "frac_without_unwritten_residuals": moved_bytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "sample_bytes_per_launch": sample_evals * bytes_per_sample / K, "history_bytes_per_launch": hist_bytes / K,
                         "valu_busy": valu_busy, "wave_wait_frac": wait_frac, "mfma": mfma,
                         # an fp64 vector instruction is priced at 4 cycles above; this part sustains 58.7 of the 78.6 TFLOP/s that would mean (tools/micro/fma_rate.hip: 5.4 cycles per
                         # fp64 FMA) and the quarter-rate instructions of the sample code take 16: by the measured FMA rate alone the pipes are this busy
                         "valu_busy_at_measured_fp64_issue_rate": (valu_busy * 78.6432 / 58.7) if valu_busy is not None else None,
                         "valu_busy_source": ("SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x launch x 2.4 GHz) of the counter pass: %s (git %s, kernel sources %s, launch %.1f ms)" % (
                             pc.get("tag"), pc.get("git_head"), pc.get("kernel_src_sha"), pc.get("launch_ms", float("nan")))) if pc is not None else pc_why,
                         "kernel_src_sha": kernel_sources_sha()},
            # which library ran: the hash compiled into libunevenhip.so against the same hash of the tree's sources (csrc/Makefile, _lib.sources_id)
            "library": {"build_id": U._lib.build_id(), "sources_id": U._lib.sources_id(), "is_a_build_of_this_tree": U._lib.build_id() == U._lib.sources_id(),
                        "path": os.environ.get("UNEVENHIP_LIB") or os.path.join(ROOT, "uneven_planner_amd", "libunevenhip.so")},
        }
        if astar_wl:
            res["metric"] = "MINCO traj-opts/sec (batch), A*-seeded inputs"
            res["front_end"] = dict(astar_meta, mean_pieces=float(np.mean([s_["Nxy"] for s_ in opt._sizes])))
        res["converged_traj_opts_per_s"] = value * res["converged_frac"]      # solves that END converged (ret_code 0); the rest hit the ALM pass cap like the reference's
        fe_queries = extras.pop("_front_end_queries", None)
        astar = extras.pop("_astar", None)
        if "penalty_kernel" in extras:
            # north_star states its >= 70 % on the penalty kernel: that kernel's own fraction sits inside `roofline`, next to the solve kernel's
            res["roofline"]["penalty_kernel"] = {"kernel": "uph_solver_kernel<128,2,0> (uph_eval_batch: %d objective + gradient evaluations per trajectory and launch)" % extras["penalty_kernel"]["batch"]["evals_per_launch"],
                                                 "frac": extras["penalty_kernel"]["batch"]["frac"], "achieved": extras["penalty_kernel"]["batch"]["achieved_GBs"], "unit": "GB/s",
                                                 "frac_moved": extras["penalty_kernel"]["batch"]["frac_moved"],
                                                 "frac_a5_only": extras["penalty_kernel"]["batch"]["a5_only"]["frac"],
                                                 "frac_hill_trajectory_x256": extras["penalty_kernel"]["hill_x256"]["frac"],
                                                 "frac_moved_hill_trajectory_x256": extras["penalty_kernel"]["hill_x256"]["frac_moved"],
                                                 "frac_a5_only_hill_trajectory_x256": extras["penalty_kernel"]["hill_x256"]["a5_only"]["frac"],
                                                 "definitions": "frac: innerCallback launches (generate + expand + samples + scatter + adjoint) priced at SURVEY.md 8d's 376 B per sample-evaluation; "
                                                                "frac_moved: the same launches by the bytes they move (the 7 x 8 B residual stores happen once per launch, not per evaluation); "
                                                                "frac_a5_only: calConstrainCostGrad alone (uph_penalty_batch, uph_solver_kernel<128,2,8>: samples + scatter, residuals stored by every call) at 376 B, all of them moved",
                                                 "target": 0.70}
        res.update(extras)
        if world == 1 and not km2 and not args.no_configs and not args.no_extras and not args.no_cpu:
            # every BASELINE.json config in this one line (outside the timed region; N = 1 only -- for N > 1 the other ranks would wait at the barrier)
            res["configs"] = measure_configs(args, torch, U, scenes, m, res, extras.get("penalty_kernel"))
        if pipelined:
            res["pipelined"] = pipelined
        if world == 1 and not args.no_cpu and args.cpu_sample > 0:
            from oracle import oracle_py as O
            nsamp = min(args.cpu_sample, len(probs))
            if not km2:
                og = O.OracleGrid()
                og.set_cells(m.map_buffer)
            cdt, c_iters, oref = 0.0, 0, []
            for p in probs[:nsamp]:
                # km2: the 1e9-cell grid stays on the device; the problem is solved on the window of cells around it, translated by whole
                # cells (window download not timed)
                g_, q_, shift = O.window_oracle(m, p) if km2 else (og, p, (0.0, 0.0))
                t0 = time.perf_counter()
                r = O.OracleALM(g_).optimize(q_)
                cdt += time.perf_counter() - t0
                c_iters += r["lbfgs_iters"]
                if km2:                       # the window's frame is the map's shifted by whole cells: way-points back into map coordinates
                    nin = np.asarray(p["inner_xy"]).reshape(2, -1).shape[1]
                    r = dict(r, x=np.array(r["x"], dtype=np.float64).copy())
                    r["x"][1:1 + 2 * nin:2] += shift[0]
                    r["x"][2:2 + 2 * nin:2] += shift[1]
                oref.append(r)
            res["cpu_baseline"] = {"value": nsamp / cdt, "unit": "traj-opts/s", "cores": 1, "kind": "port",
                                   "sample": "first %d problems of the same batch, CPU oracle (C++ -O3, single thread), %.1f s" % (nsamp, cdt),
                                   "ms_per_lbfgs_iter": cdt * 1e3 / max(1, c_iters), "cpu_model": cpu_model(), "host_cpus": os.cpu_count(),
                                   "converged_frac": float(np.mean([r_["ret"] == 0 for r_ in oref]))}
            # the headline's parity statement: the timed batch's own results (same inputs, rho reset every step) against the oracle on that sample,
            # bucketed by the oracle's L-BFGS iteration count (the optimiser amplifies rounding noise ~1.25x per iteration,
            # so <= 1e-4 on final way-points holds for short solves and decays with length -- for the oracle against its own FMA rebuild as well)
            edges = [0, 80, 120, 180, 260, 400, 1 << 30]
            rows = []
            devs = out[:nsamp]
            relx = np.array([np.abs(d_["x"] - r_["x"]).max() / max(1e-300, np.abs(r_["x"]).max()) for d_, r_ in zip(devs, oref)])
            its = np.array([r_["lbfgs_iters"] for r_ in oref])
            for lo_, hi_ in zip(edges[:-1], edges[1:]):
                sel = (its >= lo_) & (its < hi_)
                if sel.any():
                    rows.append({"iters": [lo_, hi_ if hi_ < (1 << 30) else None], "n": int(sel.sum()), "waypoints_le_1e-4": float((relx[sel] <= 1e-4).mean()), "median": float(np.median(relx[sel]))})
            drift = None
            if not km2:
                # the oracle's own reproducibility on the same sample (rebuilt with -ffp-contract=fast -march=native, tests/sensitivity.py), so that the
                # line shows all three converged rates, the discordant pairs BOTH ways and a sign test on the final costs: a one-sided drift would show
                try:
                    sys.path.insert(0, os.path.join(ROOT, "tests"))
                    import sensitivity
                    fma = sensitivity.solve_with_fma_oracle(m.map_buffer, probs[:nsamp])
                    drift = sensitivity.drift_stats(oref, fma, devs)
                except Exception as e:      # (no compiler on the box: the rest of the line stands)
                    drift = {"error": repr(e)}
            res["parity_floor"] = {"sample": nsamp, "same_ret": float(np.mean([d_["ret"] == r_["ret"] for d_, r_ in zip(devs, oref)])), "drift": drift,
                                   "waypoints_le_1e-4": float((relx <= 1e-4).mean()), "converged_frac_device_on_sample": float(np.mean([d_["ret"] == 0 for d_ in devs])),
                                   "by_oracle_lbfgs_iters": rows,
                                   "note": "device vs CPU oracle, final way-points, relative inf-norm; the oracle rebuilt with FMA contraction shows the same decay (profiles/*parity_buckets.json)" +
                                           ("; km2: the 1e9-cell grid stays on the device, the oracle runs on a window of cells translated to its own origin; the device solves every "
                                            "trajectory in a local frame a whole number of cells from the map's (uph_common.hpp TrajFrame), so both form the lookups' differences -- and the "
                                            "||x||-normalised gradient test of lbfgs.hpp:599-606 -- on numbers of the path's own size (DESIGN.md 4a)" if km2 else "")}
            if astar is not None and "astar_seeded" in res:
                # the A*-seeded batch against the oracle on its first problems: same table as the headline's parity_floor
                na = min(len(astar[0]), max(32, nsamp // 2))
                t0 = time.perf_counter()
                aref = [O.OracleALM(og).optimize(p_) for p_ in astar[0][:na]]
                adt = time.perf_counter() - t0
                adev = astar[1][:na]
                arel = np.array([np.abs(d_["x"] - r_["x"]).max() / max(1e-300, np.abs(r_["x"]).max()) for d_, r_ in zip(adev, aref)])
                aits = np.array([r_["lbfgs_iters"] for r_ in aref])
                arows = []
                for lo_, hi_ in zip(edges[:-1], edges[1:]):
                    sel = (aits >= lo_) & (aits < hi_)
                    if sel.any():
                        arows.append({"iters": [lo_, hi_ if hi_ < (1 << 30) else None], "n": int(sel.sum()), "waypoints_le_1e-4": float((arel[sel] <= 1e-4).mean()), "median": float(np.median(arel[sel]))})
                adrift = None
                try:
                    afma = sensitivity.solve_with_fma_oracle(m.map_buffer, astar[0][:na])
                    adrift = sensitivity.drift_stats(aref, afma, adev)
                except Exception as e:
                    adrift = {"error": repr(e)}
                res["astar_seeded"]["parity_floor"] = {"sample": na, "same_ret": float(np.mean([d_["ret"] == r_["ret"] for d_, r_ in zip(adev, aref)])), "waypoints_le_1e-4": float((arel <= 1e-4).mean()),
                                                       "converged_frac_oracle_on_sample": float(np.mean([r_["ret"] == 0 for r_ in aref])), "converged_frac_device_on_sample": float(np.mean([d_["ret"] == 0 for d_ in adev])),
                                                       "by_oracle_lbfgs_iters": arows, "drift": adrift}
                res["astar_seeded"]["cpu_baseline"] = {"value": na / adt, "unit": "traj-opts/s", "cores": 1, "kind": "port", "sample": "first %d problems of the A*-seeded batch, CPU oracle, %.1f s" % (na, adt)}
            if fe_queries is not None and "front_end" in res and "error" not in res["front_end"]:
                # the front end on the host: the oracle's restatement of KinoAstar::plan, single thread, first 64 queries of the same list
                og.set_occ(m.occ_buffer, m.occ_r2_buffer)
                ok_ = O.OracleKinoAstar(og)
                t0 = time.perf_counter()
                fr = [ok_.plan(s_, g_) for s_, g_ in zip(fe_queries[0][:64], fe_queries[1][:64])]
                fdt = time.perf_counter() - t0
                res["front_end"]["cpu"] = {"ms_per_goal": fdt * 1e3 / 64, "queries_per_s": 64 / fdt, "cores": 1, "kind": "port", "sample": "first 64 queries, CPU oracle (oracle/kino_astar.hpp), %.2f s" % fdt,
                                           "expansions_per_query": float(np.mean([q_["iter_num"] for q_ in fr])), "found": float(np.mean([q_["status"] == 0 for q_ in fr]))}
                big = max(int(k_[1:]) for k_ in res["front_end"] if k_.startswith("B"))
                res["front_end"]["gpu_over_cpu_per_goal"] = res["front_end"]["B%d" % big]["queries_per_s"] / (64 / fdt)
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
