#!/usr/bin/env python3
"""A 30-second first-contact probe for a multi-GPU node: before anything is measured on N GPUs, find out -- with a hard wall-clock
bound per phase, so that it can never hang a lease -- whether the pieces the N-rank engine relies on work between THESE devices:

  phase 1  rccl_init     psgsdf_create + psgsdf_comm_init on every rank (ncclCommInitRank over the engine's own dlopen'ed librccl)
  phase 2  allreduce     one collective through that communicator (the cut negotiation of a 32^3 volume: psgsdf_plan_slab)
  phase 3  hand_off      the in-kernel hand-off forms between every neighbour pair, once per memory kind (PSGSDF_XR_MEM=fine, uncached): write-through
                         records + tagged flag against a reader holding stale copies (comm.hip xr_probe), the nonce check of every IPC mapping; per rank:
                         stale records, expired waits towards the lower / upper neighbour
  phase 4  exchanges     two Gauss-Newton iterations of a 64^3 scene cut into N slabs with every in-kernel exchange on (cross-rank persistent solve, frame
                         rows, scalar folds, halo push / pull; in-kernel waits shortened to 2^20 polls ~ 1 s), against one context on the rank's own device

One JSON line on stdout (rank 0's parent): per phase the seconds every rank took, per rank the probe's view of its pairs, `ok` and -- if a phase ran
into its bound -- `timed_out_phase` with what the ranks had reported until then.  Exit code 0 only if everything passed.

    python tools/first_contact.py --gpus 8 [--phase-timeout 60]          (or: python bench.py --probe-only --gpus 8)
    PSGSDF_BENCH_SHARE_GPU=1 python tools/first_contact.py --gpus 4      (all ranks on GPU 0 through the gloo test transport: a functional check of the tool)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PHASES = ["rccl_init", "allreduce", "hand_off", "exchanges"]


def worker(rank, world, out_path, share):
    import numpy as np
    import torch
    import torch.distributed as dist
    from psgradientsdf_amd import capi, synth
    os.environ.setdefault("PSGSDF_XWAIT_LOG2", "20")      # in-kernel waits for a peer: ~1 s instead of ~20 s -- a lost peer shows at once
    log = open(out_path, "a", buffering=1)

    def report(phase, t0, **kw):
        log.write(json.dumps({"rank": rank, "phase": phase, "seconds": round(time.time() - t0, 3), **kw}) + "\n")

    device = 0 if share else rank
    torch.cuda.set_device(device)
    if share:
        ncu = torch.cuda.get_device_properties(0).multi_processor_count
        os.environ.setdefault("PSGSDF_CU_MASK", f"{rank * (ncu // world)}:{(rank + 1) * (ncu // world)}")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    # (torch's process group only hands the RCCL id around and lines the ranks up between the phases)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    def attach(eng):
        if share:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from _gloo_transport import GlooTransport
            eng._transport = GlooTransport(dist)
            eng.comm_init_ext(eng._transport.ops, rank, world)
        else:
            ident = [capi.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ident, src=0)
            eng.comm_init(rank, world, ident[0])

    # ---- phase 1 + 2: communicator, one collective
    sc = synth.make_scene(N=32, F=4, W=96, H=72, model="SH1")
    st = capi.default_settings(capi.SH1)
    t0 = time.time()
    eng = capi.load_engine(sc, sc.K, st, device)
    attach(eng)
    report("rccl_init", t0, transport="gloo test transport (shared GPU)" if share else "rccl")
    dist.barrier()
    t0 = time.time()
    nz = int(sc.dim[2]); plane = int(sc.dim[0]) * int(sc.dim[1])
    cnt = np.zeros(nz)
    for k in range(rank, nz, world):
        cnt[k] = eng.slab_plane_count(sc.dist[k * plane:(k + 1) * plane], sc.vis[k * plane:(k + 1) * plane], sc.vis_words)
    z0, z1 = eng.plan_slab(cnt)                       # ONE all-reduce of the per-plane histogram through the engine's communicator
    report("allreduce", t0, planes=[z0, z1])
    eng.close()
    dist.barrier()
    # ---- phase 3: the hand-off forms per memory kind, between the real neighbours
    sc = synth.make_scene(N=64, F=8, W=160, H=120, model="SH1")
    t0 = time.time()
    kinds = {}
    for kind in ("fine", "uncached"):
        os.environ["PSGSDF_XR_MEM"] = kind
        eng = capi.load_engine(sc, sc.K, st, device)
        attach(eng)
        eng.load_scene_slab(sc, rank, world)          # the first band of a multi-rank context runs the probe and the nonce check
        tu = eng.get_tuning().get("xr_probe", {})
        ss = eng.debug_sync_stats()
        kinds[kind] = {"passed_on_all_ranks": int(ss["cross_rank_mem_kind"]) == (1 if kind == "fine" else 2), "cross_rank_ready": int(ss["cross_rank_ready"]),
                       "this_rank": tu.get("fine_grained" if kind == "fine" else "uncached"), "stale_mappings": tu.get("stale_mappings"),
                       "all_ranks_stale_records": int(ss["probe_stale"]), "all_ranks_expired_waits": int(ss["probe_timeouts"])}
        eng.close()
        dist.barrier()
    del os.environ["PSGSDF_XR_MEM"]
    report("hand_off", t0, neighbours=[rank - 1 if rank > 0 else None, rank + 1 if rank < world - 1 else None], kinds=kinds)
    # ---- phase 4: every in-kernel exchange in two real iterations, against one context
    t0 = time.time()
    eng = capi.load_engine(sc, sc.K, st, device)
    attach(eng)
    eng.load_scene_slab(sc, rank, world)
    eng.init_albedo(); eng.normalize_weights()
    err = None
    try:
        e_n = [float(r["e_total"]) for r in eng.iterate(capi.ALL, 2)]
    except capi.PsgsdfError as ex:
        e_n, err = None, str(ex)[-400:]
    ss = eng.debug_sync_stats(); ncoll = eng.comm_stats()
    eng.close()
    ref = capi.load_engine(sc, sc.K, st, device)
    ref.load_scene(sc); ref.init_albedo(); ref.normalize_weights()
    e_1 = [float(r["e_total"]) for r in ref.iterate(capi.ALL, 2)]
    ref.close()
    rel = max(abs(a - b) / abs(b) for a, b in zip(e_n, e_1)) if e_n else None
    report("exchanges", t0, error=err, e_total_rel_diff=rel, ok=bool(rel is not None and rel <= 1e-5), cross_rank_ready=int(ss["cross_rank_ready"]), cross_rank_solves=int(ss["cross_rank_solves"]),
           persist_fallbacks=int(ss["persist_fallbacks"]), halo_exchanges_by_push_kernels=int(ss["halo_pushes"]), hand_off_memory=int(ss["cross_rank_mem_kind"]), communicator_calls=int(ncoll))
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=2)
    ap.add_argument("--phase-timeout", type=float, default=60.0, help="wall-clock bound of every phase (seconds); the first phase also pays the interpreter / torch start-up")
    ap.add_argument("--worker", nargs=3)
    a = ap.parse_args()
    share = os.environ.get("PSGSDF_BENCH_SHARE_GPU") == "1"
    if a.worker:
        worker(int(a.worker[0]), int(a.worker[1]), a.worker[2], share)
        return 0
    import torch
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    n = a.gpus
    if (ndev < n and not share) or ndev < 1:
        print(json.dumps({"ok": False, "error": f"--gpus {n} needs {n} devices, {ndev} visible (PSGSDF_BENCH_SHARE_GPU=1: all ranks on GPU 0, a functional check of the tool)"}))
        return 3
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    td = tempfile.mkdtemp(prefix="first_contact_")
    procs, logs = [], []
    t_start = time.time()
    for r in range(n):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", PSGSDF_DESTROY_TIMEOUT_S="3")
        logs.append(os.path.join(td, f"rank{r}.jsonl")); open(logs[-1], "w").close()
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--gpus", str(n), "--worker", str(r), str(n), logs[-1]], env=env,
                                      stdout=subprocess.DEVNULL, stderr=open(os.path.join(td, f"rank{r}.err"), "w")))

    def read():
        recs = []
        for p in logs:
            for line in open(p):
                try:
                    recs.append(json.loads(line))
                except ValueError:
                    pass
        return recs

    timed_out, phase_i, t_phase = None, 0, time.time()
    budget = a.phase_timeout + 90.0                     # (phase 1 includes python + torch start-up on a fresh box)
    while phase_i < len(PHASES):
        recs = read()
        done = sum(1 for x in recs if x["phase"] == PHASES[phase_i])
        if done == n:
            phase_i += 1; t_phase = time.time(); budget = a.phase_timeout
            continue
        if any(p.poll() not in (None, 0) for p in procs):
            timed_out = f"{PHASES[phase_i]} (a rank exited with an error)"
            break
        if time.time() - t_phase > budget:
            timed_out = PHASES[phase_i]
            break
        time.sleep(0.1)
    for p in procs:                                     # exactly the processes started here
        if p.poll() is None:
            if timed_out:
                p.terminate()
    for p in procs:
        try:
            p.wait(timeout=15)
        except subprocess.TimeoutExpired:
            p.kill()
    recs = read()
    out = {"tool": "first_contact", "ranks": n, "transport": "gloo test transport, all ranks on GPU 0" if share else "rccl", "wall_s": round(time.time() - t_start, 1), "phases": {}}
    for ph in PHASES:
        rs = sorted((x for x in recs if x["phase"] == ph), key=lambda x: x["rank"])
        if rs:
            out["phases"][ph] = {"ranks_reported": len(rs), "seconds_max": max(x["seconds"] for x in rs), "per_rank": [{k: v for k, v in x.items() if k not in ("phase",)} for x in rs]}
    ex = out["phases"].get("exchanges", {}).get("per_rank", [])
    ho = out["phases"].get("hand_off", {}).get("per_rank", [])
    out["hand_off_memory_kinds_that_pass"] = [k for k in ("fine", "uncached") if ho and all(x["kinds"][k]["passed_on_all_ranks"] for x in ho)]
    out["timed_out_phase"] = timed_out
    if timed_out:
        out["stderr_tails"] = {r: open(os.path.join(td, f"rank{r}.err")).read()[-600:] for r in range(n)}
    out["ok"] = bool(not timed_out and len(ex) == n and all(x["ok"] and x["cross_rank_ready"] == 1 and x["persist_fallbacks"] == 0 for x in ex) and out["hand_off_memory_kinds_that_pass"])
    print(json.dumps(out), flush=True)
    return 0 if out["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
