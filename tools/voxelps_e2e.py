#!/usr/bin/env python3
"""End-to-end wall-clock of the voxelPS drop-in, per stage: config_skorates.json's settings on the reference's demo data --
(a) the four native-resolution frames (1139 x 1709, tests/golden/sokrates_native_4), (b) the 21 sub-sampled frames (380 x 570, tests/golden/sokrates_21).
Runs `voxelPS --config_file .. --timing ..` and collects the stage table; `--exe` / `--label` allow before / after comparisons of two builds.

    python tools/voxelps_e2e.py [--out gpurun_out/voxelps_e2e.json] [--label after] [--reps 2]
"""
import argparse, json, os, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = {"native_4": ("tests/golden/sokrates_native_4", 3), "frames_21": ("tests/golden/sokrates_21", 20)}


def run(exe, name, reps, extra_flags=()):
    inp, last = DATA[name]
    best = None
    for rep in range(reps):
        with tempfile.TemporaryDirectory() as td:
            out = td + "/"
            cfg = {"input": os.path.join(ROOT, inp) + "/", "output": out, "pose filename": "pose.txt", "datatype": "multiview", "first": 0, "last": last, "voxel size": 0.004,
                   "truncation factor": 5, "zmin": 0.5, "zmax": 3.5, "sharpness threshold": 0.0, "model type": "SH1", "loss function": "cauchy",
                   "reg albedo": 0.0, "reg norm": 10.0, "reg laplacian": 0.0, "max iter": 100, "damping": 1.0, "converge threshold": 5e-3, "lambda": 0.2,
                   "upsample": False, "--light": True, "--albedo": True, "--distance": True, "--pose": True}      # config_skorates.json
            json.dump(cfg, open(out + "config.json", "w"))
            t0 = time.time()
            r = subprocess.run([exe, "--config_file", out + "config.json", "--timing", out + "timing.json"] + list(extra_flags), capture_output=True, text=True, timeout=900)
            wall = time.time() - t0
            if r.returncode != 0:
                return {"error": r.stdout[-500:] + r.stderr[-500:]}
            t = json.load(open(out + "timing.json"))
            files = {f: os.path.getsize(out + f) for f in sorted(os.listdir(out)) if f.endswith((".ply", ".sdf", ".txt"))}
            iters = r.stdout.count("relative diff")
            t.update(wall_s=wall, iterations=iters, output_bytes=sum(files.values()), n_files=len(files), ended=("converged" if "converged!" in r.stdout else "diverged" if "diverged!" in r.stdout else "max iter"))
            if best is None or t["total_s"] < best["total_s"]:
                best = t
    s = best["stages_s"]
    get = lambda pre: sum(v for k, v in s.items() if k.startswith(pre))
    best["summary_s"] = {"decode (main thread)": get("decode:"), "fuse": get("fuse:"), "focus_measure": get("keyframe selection"), "dumps (main thread)": get("dump:"),
                         "background writer (overlapped)": get("background:"), "waiting for the writer": get("wait for the background"),
                         "alternatingOptimize incl. its dumps": get("alternatingOptimize"), "total": best["total_s"]}
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--exe", default=os.path.join(ROOT, "psgradientsdf_amd", "host", "voxelPS"))
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "voxelps_e2e.json"))
    ap.add_argument("--label", default="run")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--host-writers", action="store_true", help="round 4's path: dense downloads, host marching cubes, iostream, serial decode")
    a = ap.parse_args()
    res = {"label": a.label, "exe": os.path.relpath(a.exe, ROOT)}
    for name in DATA:
        res[name] = run(a.exe, name, a.reps, ["--host-writers"] if a.host_writers else [])
        print(name, json.dumps(res[name]), flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    old = json.load(open(a.out)) if os.path.exists(a.out) else {}
    old[a.label] = res
    json.dump(old, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
