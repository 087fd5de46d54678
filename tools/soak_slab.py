"""Reproducibility soak of the MULTI-RANK loop on one GPU: world-size-N runs of the engine's native slab loop (tests/_slab_worker_gpu.py: N processes
sharing GPU 0 through the gloo test transport, the in-kernel exchanges over IPC mappings between them) repeated `--reps` times per configuration;
every repetition must reproduce the first one bit for bit (distances, albedo, poses, light, energies, CG iteration counts) on every rank.

    python tools/soak_slab.py --reps 10 --out gpurun_out/soak_slab.json
"""
import argparse, hashlib, json, os, socket, subprocess, sys, tempfile, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EIGHTHS = {"SLAB_CU_MASKS": ",".join(f"{32 * i}:{32 * i + 32}" for i in range(8)), "SLAB_WORKER_TIMEOUT": "190"}
CONFIGS = [("SH1", 2, "optimize", 24, 18, {}), ("LED", 3, "optimize", 24, 18, {}), ("SH1", 4, "iterate", 40, 2, {}), ("SH2", 2, "iterate", 40, 2, {}), ("SH1", 3, "refine", 24, 2, {}),
           # world size 8: eight ranks on CU eighths, cross-rank persistent solve on; SH2 with two visibility words; the loop through the refinement
           ("SH1", 8, "iterate", 64, 2, EIGHTHS), ("SH2", 8, "iterate", 48, 2, dict(EIGHTHS, SLAB_FRAMES="70:96:72")), ("SH1", 8, "optimize", 24, 18, EIGHTHS)]


def run(model, world, mode, N, n_iters, out, extra_env=None):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GLOO_SOCKET_IFNAME="lo", **(extra_env or {}))
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_slab_worker_gpu.py"), str(r), str(world), str(port), model, out, str(n_iters), str(N), "gloo", mode],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(world)]
    err = None
    try:
        for p in procs:
            o, _ = p.communicate(timeout=200)
            if p.returncode != 0:
                err = o[-600:]
    except subprocess.TimeoutExpired:
        err = "timeout"
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    if err:
        return None, err
    sig = []
    for r in range(world):
        z = np.load(out + f".rank{r}.npz")
        h = hashlib.sha1()
        for k in ("dist", "rgb", "poses", "light", "e_total", "cg"):
            h.update(np.ascontiguousarray(z[k]).tobytes())
        sig.append(h.hexdigest()[:16])
        stats = [int(x) for x in z["xr"]] + [int(z["halo_pushes"])]
    return sig, stats


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "soak_slab.json"))
    ap.add_argument("--ranks", type=int, default=0, help="only the configurations with this many ranks (0: all)")
    a = ap.parse_args()
    report, bad = [], 0
    t0 = time.time()
    with tempfile.TemporaryDirectory() as td:
        for (model, world, mode, N, it, env) in CONFIGS:
            if a.ranks and world != a.ranks:
                continue
            sigs, errs, stats = [], [], None
            for rep in range(a.reps):
                sig, st = run(model, world, mode, N, it, os.path.join(td, f"{model}{world}{mode}{rep}"), env)
                if sig is None:
                    errs.append({"rep": rep, "err": st}); continue
                sigs.append(tuple(sig)); stats = st
            distinct = len(set(sigs))
            bad += len(errs) + (distinct > 1)
            report.append(dict(model=model, ranks=world, mode=mode, grid=N, runs=a.reps, completed=len(sigs), distinct_results=distinct, errors=errs,
                               last_run_stats=dict(zip(("cross_rank_ready", "cross_rank_solves", "persist_fallbacks", "mem_kind", "probe_stale", "probe_timeouts", "halo_pushes"), stats or []))))
            print(report[-1], flush=True)
    json.dump(dict(wall_s=round(time.time() - t0, 1), deviating=bad, configs=report), open(a.out, "w"), indent=1)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
