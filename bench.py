#!/usr/bin/env python3
"""bench.py — Gauss-Newton iterations/sec of the full photometric-stereo sweep (BASELINE.json metric).

One "step" = one body of the alternation loop (albedo, light, distance, pose blocks, the four PS-energy
evaluations and the convergence test; PsOptimizer.cpp:303-366) on the synthetic 256^3 x 50-keyframe
scene, inputs resident in HBM before the timed region.  Prints ONE JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--grid 256] [--frames 50] [--model SH1]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: weak scaling over z-slabs.  The grid grows to 256 x 256 x (256*N) -- N copies of the scene stacked
along z, each with its own 50 keyframes -- one process per GPU owns one slab and calls the SAME psgsdf_iterate: the engine's C++ host
exchanges halos / all-reduces over its own RCCL communicator (comm.hip).  Python only launches the ranks, hands the RCCL id around and
takes the time.  value = N * it/s of the whole job (256^3 x 50-frame equivalents per second).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from psgradientsdf_amd import capi, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes(kernel, S, n_obs, W, H, F, laplacian=False, pcg_passes=1.0):
    """Algorithmic HBM bytes of ONE launch: per band voxel B_v = 60 B of state
    (dist 4, grad 12, rgb 12, vis 8, 3 stencil-neighbour dists 12, 3 row lookups 12; +12 with the Laplacian),
    image taps U_img = min(48 B * n_obs, 12 B * W*H*F) and the kernel's per-voxel output."""
    B_v = 60 + (12 if laplacian else 0) + (8 if F > 64 else 0)
    U = min(48 * n_obs, 12 * W * H * F)
    table = {
        "sweep_albedo": S * B_v + U + 24 * S,
        "sweep_light": S * B_v + U,
        "sweep_pose": S * B_v + U,
        "sweep_dist": S * B_v + U + 56 * S,
        "energy": S * B_v + U,
        "pcg_solve": 124 * S * pcg_passes,   # the persistent solve: ONE launch runs all passes of a solve (B_cg = 124 B per band voxel per PCG iteration)
        "pcg_pass": 124 * S,    # B_cg = 124 B per band voxel per PCG iteration (block 40 + 3 nbr rows 12 + gather p 16 + scatter Ap 16 + 5 vector streams 40)
        "assemble": (56 + 80 + 12) * S,
        "derive": (4 + 12 + 24 + 36 + 12) * S,
    }
    return table.get(kernel)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--frames", type=int, default=50)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--model", default="SH1", choices=["SH1", "SH2", "LED"])
    ap.add_argument("--u8-images", action="store_true", help="keyframes quantised to 8 bits and handed over as 8-bit RGB (psgsdf_set_keyframes_u8) instead of float RGB")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true")
    ap.add_argument("--force-slab", action="store_true", help="attach the single rank to a one-rank RCCL communicator (overhead of the multi-rank code path)")
    args = ap.parse_args()

    if os.environ.get("PSGSDF_FAULT_DUMP"):   # diagnostics: dump every thread's Python stack after N seconds and exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["PSGSDF_FAULT_DUMP"]), exit=True)
    # RCCL / HIP runtime banners go to the C-level stdout: keep the real stdout for the ONE JSON line only
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    import torch
    slab = world > 1 or args.force_slab
    share = False
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        # PSGSDF_BENCH_SHARE_GPU=1: every rank on GPU 0 and the exchanges through the gloo test transport (RCCL refuses two ranks per
        # device) -- a functional check of this code path on a one-GPU box (tests/test_bench_gpu.py), not a measurement
        share = os.environ.get("PSGSDF_BENCH_SHARE_GPU") == "1"
        if share:
            local_rank = 0
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        torch.cuda.set_device(local_rank)
        if share:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:      # torch's process group only carries the RCCL id, the barriers and the MAX of the elapsed time
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
    device = local_rank if world > 1 else 0

    t_gen = time.time()
    # --u8-images: keyframes quantised to 8 bits and handed over the way the reference's loader receives them (8-bit RGB + 1/255,
    # ImageLoader.h:167-181); the oracle always gets the converted floats.  ~3 % faster than the float path (profiles/r01_notes.md, step p).
    use_u8 = args.u8_images
    sc = synth.make_scene(N=args.grid, F=args.frames, W=args.width, H=args.height, model=args.model, u8=use_u8)
    t_gen = time.time() - t_gen
    model_id = synth.MODELS[args.model]
    st = capi.default_settings(model_id)
    if args.model == "LED":   # config_basket_LED.json
        st.reg_weight_n, st.reg_weight_l, st.damping = 0.1, 5.0, 3.0
    if world > 1:
        sc = synth.tile_scene(sc, world)
    eng = capi.load_engine(sc, sc.K, st, device)
    transport = None
    if slab:
        if share:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from _gloo_transport import GlooTransport
            transport = GlooTransport(dist)
            eng.comm_init_ext(transport.ops, rank, world)
        else:
            ident = [capi.comm_unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(ident, src=0)
            eng.comm_init(rank, world, ident[0])
    eng.load_scene(sc, u8=use_u8)
    eng.init_albedo()
    eng.normalize_weights()
    S = eng.info().n_band // world              # per-slab band size
    n_obs = eng.step(capi.ALBEDO)["n_obs"] // world
    iterate = lambda k: eng.iterate(capi.ALL, k)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    iterate(args.warmup)
    # pick the dominant kernel (largest total time per iteration) from a short synchronous-event pass, then time THAT
    # kernel inside the timed region with HIP events recorded on the launch stream (no host sync)
    kernels, dom = {}, "sweep_dist"
    if not args.no_breakdown:
        eng.set_profiling(True)
        eng.reset_kernel_times()
        nprof = 3
        iterate(nprof)
        kt = eng.kernel_times()
        eng.set_profiling(False)
        kernels = {k: {"ms_per_iter": v[0] / nprof, "launches_per_iter": v[1] / nprof} for k, v in kt.items()}
        dom = max(kernels, key=lambda k: kernels[k]["ms_per_iter"] if algorithmic_bytes(k, 1, 1, 1, 1, 1) else -1)
    eng.reset_kernel_times()
    if os.environ.get("PSGSDF_NO_WATCH") != "1":   # (tools/gap_run.sh: trace without the event pairs)
        eng.watch_kernel(dom + ("/16" if dom == "pcg_pass" else "/4"))  # HIP events around every 16th (4th) launch of the dominant kernel (each pair breaks the back-to-back dispatch: ~6 us of stream time)
    barrier()
    coll0 = eng.comm_stats()
    t0 = time.perf_counter()
    recs = iterate(args.steps)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    coll1 = eng.comm_stats()
    barrier()
    elapsed = t1 - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{device}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    watched = eng.kernel_times().get(dom, (0.0, 0))
    eng.watch_kernel("")
    # the loop voxelPS runs (psgsdf_optimize: stop decision after every iteration) on the same state, next to psgsdf_iterate's figure
    opt_ms = None
    if world == 1 and not slab:
        st2 = capi.default_settings(model_id)
        st2.reg_weight_n, st2.reg_weight_l, st2.damping = st.reg_weight_n, st.reg_weight_l, st.damping
        st2.max_it, st2.conv_threshold, st2.upsample = args.steps, 0.0, 0
        eng2 = capi.load_engine(sc, sc.K, st2, device)
        eng2.load_scene(sc, u8=use_u8)
        torch.cuda.synchronize(); to = time.perf_counter()      # (kernels are loaded: the first context ran them all)
        recs2, _ = eng2.optimize(capi.ALL)              # initAlbedo + weight normalisation + up to `steps` iterations, as PsOptimizer::alternatingOptimize
        torch.cuda.synchronize(); to = time.perf_counter() - to
        opt_ms = 1e3 * to / max(len(recs2), 1)
        eng2.close()

    ms_per_step = 1e3 * elapsed / args.steps
    value = world * args.steps / elapsed
    cg_iters = float(np.mean([r["cg_iters"] for r in recs]))

    out = {
        "metric": "Gauss-Newton iterations/sec (full PS sweep), 256^3 grid x 50 frames" if (args.grid, args.frames) == (256, 50)
        else f"Gauss-Newton iterations/sec (full PS sweep), {args.grid}^3 grid x {args.frames} frames",
        "value": value, "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic (8-bit RGB keyframes)" if use_u8 else "synthetic",
        "config": {"workload": f"synthetic {args.width}x{args.height} RGB-D bumpy sphere, {args.grid}^3 grid, {args.model}, {args.frames} keyframes, "
                               "albedo+light+distance+pose blocks, Cauchy IRLS, Eikonal reg (config_skorates.json settings)",
                   "band_voxels": int(S), "observations": int(n_obs), "pcg_iters_per_step": cg_iters,
                   "parallelism": "single GPU" if world == 1 else f"{world} z-slabs (one per GPU), grid {args.grid}x{args.grid}x{args.grid * world}, {args.frames * world} keyframes, native slab loop: RCCL halo exchange + all-reduce issued by the C++ host"},
    }

    if rank == 0:
        # ---- roofline of the dominant kernel, timed live with HIP events inside the timed region
        lap = st.reg_weight_l != 0.0
        # event pairs recorded on the launch stream around every 16th launch: the pair also reads the dispatch latency, so this is
        # ~2.5 us above the kernel duration rocprofv3 --kernel-trace reports (profiles/): the roofline fraction errs low
        avg_ms = watched[0] / max(watched[1], 1) if watched[1] else float("nan")
        nbytes = algorithmic_bytes(dom, S, n_obs, args.width, args.height, args.frames, lap, pcg_passes=cg_iters + 1.0)   # per GPU (one slab); a solve of n iterations runs n + 1 passes
        # the solve enqueues the previous solve's pass count + 2: the surplus kernels return at once (no-op launches, ~4.7 us) and are part
        # of the sampled average; `avg_working_launch_ms` removes them with the measured ratio passes / launches
        launches_per_step = kernels.get(dom, {}).get("launches_per_iter") if kernels else None
        working_frac = min(1.0, (cg_iters + 1.0) / launches_per_step) if (dom == "pcg_pass" and launches_per_step) else 1.0
        noop_ms = 0.0047
        avg_work_ms = (avg_ms - (1.0 - working_frac) * noop_ms) / working_frac if working_frac > 0 else avg_ms
        achieved = nbytes / (avg_ms * 1e-3) / 1e9
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if os.path.exists(pmc) and (args.grid, args.frames, args.model, world) == (256, 50, "SH1", 1):   # the counters were collected on this configuration
            try:
                pj = json.load(open(pmc))
                traffic = (pj.get(dom) or {}).get("hbm_bytes_per_launch")
                traffic_src = f"static: profiles/pmc_summary.json @{pj.get('commit', '?')} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, FETCH_SIZE x2 (gfx950), KiB -> bytes); not re-measured in this run"
            except Exception:
                traffic = None
        out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "bytes_per_launch": nbytes, "avg_launch_ms": avg_ms,
                           "launches_timed": int(watched[1]), "avg_working_launch_ms": avg_work_ms, "working_launch_fraction": working_frac,
                           "bytes_per_unit": "SURVEY 8d algorithmic figure (PCG: 124 B per band voxel per pass; pcg_solve runs cg_iters + 1 passes per launch and keeps the matrix on chip: its HBM-side traffic is far below the algorithmic bytes; the launch also assembles the distance system from the sweep's voxel blocks, which is NOT counted here)",
                           "storage_bytes_per_launch": 152 * S if dom in ("pcg_pass", "pcg_solve") else None,
                           # the pcg_solve launch also assembles the system (reads the sweep's 14-float voxel blocks + 3 neighbour rows: 68 B per band voxel)
                           # and applies the distance update (8 B): the same fraction with those algorithmic bytes counted, for reference only
                           "frac_with_fused_steps": ((nbytes + 76 * S) / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if dom == "pcg_solve" else None,
                           "traffic_source": traffic_src}
        # whole-iteration algorithmic bytes (SURVEY.md §8d formula) for reference
        U = min(48 * n_obs, 12 * args.width * args.height * args.frames)
        B_iter = 4 * (S * 60 + U) + 120 * S + 124 * cg_iters * S   # SURVEY §8d per-pass figure kept (the fused pass moves 144 B/row)
        if slab:
            out["config"]["collectives_per_step"] = (coll1 - coll0) / max(args.steps, 1)
        out["iteration"] = {"algorithmic_bytes": B_iter, "achieved_GBs": B_iter * (value / world) / 1e9,
                            "frac_of_hbm_peak": B_iter * (value / world) / 1e9 / HBM_PEAK_GBS}
        if opt_ms is not None:
            out["optimize_ms_per_step"] = opt_ms      # psgsdf_optimize (host decides convergence / divergence after every iteration)
        if kernels:
            out["kernels"] = {k: round(v["ms_per_iter"], 4) for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["ms_per_iter"])}
        out["setup_s"] = {"scene_generation": round(t_gen, 1)}

        # ---- CPU baseline: the oracle (a port of the reference's arithmetic) on the host cores, bounded sample
        if world == 1 and not slab and not args.no_cpu_baseline:
            from oracle import oracle
            orc = oracle.Oracle(sc, sc.K, st, threads=1)
            orc.load_scene(sc)
            orc.init_albedo()
            orc.normalize_weights()
            tc = time.perf_counter()
            orc.iterate(capi.ALL, 1)
            tc = time.perf_counter() - tc
            out["cpu_baseline"] = {"value": 1.0 / tc, "unit": "it/s", "cores": 1, "kind": "port",
                                   "sample": f"1 full Gauss-Newton iteration of the same {args.grid}^3 x {args.frames} scene "
                                             f"(4 blocks + 4 energy evaluations), single-threaded C oracle with indexed band lookup, {tc:.1f} s"}
            orc.close()
            # the same port with its sweeps spread over the host's cores (OpenMP): informative only, `value` stays the one-core figure
            nthr = min(64, os.cpu_count() or 1)
            if nthr > 1:
                orc = oracle.Oracle(sc, sc.K, st, threads=nthr)
                orc.load_scene(sc); orc.init_albedo(); orc.normalize_weights()
                tc = time.perf_counter(); orc.iterate(capi.ALL, 1); tc = time.perf_counter() - tc
                out["cpu_baseline"]["multithreaded"] = {"value": 1.0 / tc, "unit": "it/s", "cores": nthr}
                orc.close()
        print(json.dumps(out), file=real_stdout, flush=True)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
