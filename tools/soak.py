"""Reproducibility soak of the engine.

For every model x {iterate(n), optimize} x knob variant, runs many FRESH contexts -- in `--procs` fresh processes with `--reps`
contexts each -- on one scene and records a checkpoint vector per run: per iteration the energy after every sub-step (e_after[4]),
e_total, the CG iteration count, then hashes of the final distances / albedo / poses / light.  All variants are meant to be
bit-neutral, so every run of a (model, mode) must give the same vector; the report names, per deviating run, the FIRST checkpoint
at which it leaves the majority (iteration and sub-step), its variant, process and repetition.

    python tools/soak.py --models SH1,LED,SH2 --n 64 --frames 8 --procs 20 --reps 3 --out gpurun_out/soak

The knobs (read by psgsdf_create from the environment) are the product's latency tricks and their safe counterparts:
    PSGSDF_FOLD_IN_NEXT=0   scalar folds by a kernel of their own          PSGSDF_PCG_POLL=0     drain the stream instead of watching the mailbox
    PSGSDF_PCG_PERSIST=0    per-pass PCG kernels                            PSGSDF_MBOX_CHECK=0   read-backs not validated against their check words
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = {
    "default": {},
    "fold0": {"PSGSDF_FOLD_IN_NEXT": "0"},
    "poll0": {"PSGSDF_PCG_POLL": "0"},
    "persist0": {"PSGSDF_PCG_PERSIST": "0"},     # per-pass kernels: the CLASSIC recurrences (family "classic": bit-identical among themselves, rounding-level apart from the pipelined default)
    "pipeline0": {"PSGSDF_PCG_PIPELINE": "0"},   # persistent kernel with the classic recurrences (same family as persist0)
    "tagm0": {"PSGSDF_PCG_TAGM": "0"},           # pipelined solve without the self-validating exchanged values (round 4's hand-off: tags, acquire, drain) -- family "untagged": 2^-48 of the exchanged values apart
    "prefetch0": {"PSGSDF_PCG_PREFETCH": "0"},   # pipelined solve: sums requested after the last gather batch
    "fmsolve0": {"PSGSDF_FM_SOLVE": "0"},        # light / pose solves as kernels of their own
    "fmsolve2": {"PSGSDF_FM_SOLVE": "2"},        # ... only the LED light vector
    "xcdmap0": {"PSGSDF_XCD_MAP": "0"},          # physical workgroup ids (no XCD-contiguous mapping)
    "xcdmap7": {"PSGSDF_XCD_MAP": "7"},          # ... also for the distance sweep
    "xcdmap99": {"PSGSDF_XCD_MAP": "99"},        # heaviest-first dispatch for every per-observation voxel-major kernel
    "xcdlocal0": {"PSGSDF_PCG_XCD_LOCAL": "0"},  # persistent solve: every record through memory instead of staying in the XCD's L2 where all its readers are
    "spec0": {"PSGSDF_SPECULATE": "0"},          # every iteration closed before the next one starts (round 2)
    "nocheck": {"PSGSDF_MBOX_CHECK": "0", "PSGSDF_USE_DEV_LIB": "1"},      # read-backs taken on the marker's say-so (round 2): expected to deviate now and then
}
KNOB_NAMES = sorted({k for v in VARIANTS.values() for k in v})
FAMILY = {"persist0": "classic", "pipeline0": "classic", "tagm0": "untagged"}      # which distance-solve recurrences a variant runs (default: pipelined)


def _hash(a):
    import numpy as np
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:12]


def worker(args):
    import numpy as np
    from psgradientsdf_amd import capi, synth
    sc = synth.make_scene(N=args.n, F=args.frames, W=args.width, H=args.height, model=args.model)
    st = capi.default_settings(sc.model_id)
    variants = args.variants.split(",")
    for rep in range(args.reps):
        for vname in variants:
            for k in KNOB_NAMES:
                os.environ.pop(k, None)
            os.environ.update(VARIANTS[vname])
            for mode in args.modes.split(","):
                eng = capi.load_engine(sc, sc.K, st, 0)
                eng.load_scene(sc); eng.init_albedo(); eng.normalize_weights()
                cp, labels = [], []
                try:
                    if mode == "iterate":
                        recs = eng.iterate(capi.ALL, args.iters)
                    else:
                        recs, conv = eng.optimize(capi.ALL)
                    for i, r in enumerate(recs):
                        for q in range(4):
                            cp.append(repr(float(r["e_after"][q]))); labels.append(f"it{i}.e_after[{q}]")
                        cp.append(repr(float(r["e_total"]))); labels.append(f"it{i}.e_total")
                        cp.append(repr(int(r["cg_iters"]))); labels.append(f"it{i}.cg_iters")
                    cp.append(str(len(recs))); labels.append("n_iterations")
                    v = eng.download_volume(); band = eng.download_band()
                    cp += [_hash(v["dist"][band]), _hash(v["rgb"][:, band]), _hash(eng.download_poses()), _hash(eng.download_light())]
                    labels += ["final.dist", "final.rgb", "final.poses", "final.light"]
                    err = None
                except Exception as ex:      # an engine error is a finding too
                    err = str(ex)[:300]
                try:
                    sync = eng.debug_sync_stats()
                except Exception:
                    sync = {}
                print("SOAK " + json.dumps(dict(model=args.model, mode=mode, variant=vname, proc=args.proc, rep=rep, cp=cp, labels=labels, err=err, sync=sync)), flush=True)
                eng.close()


def analyse(runs):
    report = []
    groups = {}
    for r in runs:
        groups.setdefault((r["model"], r["mode"], FAMILY.get(r["variant"], "pipelined")), []).append(r)
    bad_total = 0
    for key, rs in sorted(groups.items()):
        # majority vector, checkpoint by checkpoint (the vectors can differ in length when the iteration count differs: compare by label)
        from collections import Counter
        label_vals = {}
        for r in rs:
            for l, v in zip(r["labels"], r["cp"]):
                label_vals.setdefault(l, Counter())[v] += 1
        whole = Counter(tuple(r["cp"]) for r in rs)
        maj = whole.most_common(1)[0][0]
        ref = next(r for r in rs if tuple(r["cp"]) == maj)
        bad = []
        for r in rs:
            if r["err"]:
                bad.append(dict(variant=r["variant"], proc=r["proc"], rep=r["rep"], first="ERROR", err=r["err"])); continue
            if tuple(r["cp"]) == maj:
                continue
            first = None
            for i, (l, v) in enumerate(zip(r["labels"], r["cp"])):
                if i >= len(ref["cp"]) or ref["labels"][i] != l or ref["cp"][i] != v:
                    first = dict(label=l, got=v, majority=ref["cp"][i] if i < len(ref["cp"]) else None); break
            bad.append(dict(variant=r["variant"], proc=r["proc"], rep=r["rep"], first=first))
        bad_total += len(bad)
        by_variant = Counter(r["variant"] for r in rs)
        dev_by_variant = Counter(b["variant"] for b in bad)
        late = Counter(); checked = Counter()
        for r in rs:
            late[r["variant"]] += (r.get("sync") or {}).get("readbacks_late", 0); checked[r["variant"]] += (r.get("sync") or {}).get("readbacks_checked", 0)
        report.append(dict(model=key[0], mode=key[1], family=key[2], runs=len(rs), distinct_results=len(whole), deviating=len(bad), runs_per_variant=dict(by_variant), deviating_per_variant=dict(dev_by_variant),
                           readbacks_checked=dict(checked), readbacks_late=dict(late), deviations=bad[:40]))
    return report, bad_total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--models", default="SH1,LED,SH2")
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--width", type=int, default=160)
    ap.add_argument("--height", type=int, default=120)
    ap.add_argument("--procs", type=int, default=10)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--modes", default="iterate,optimize")
    ap.add_argument("--variants", default=",".join(VARIANTS))
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "soak"))
    ap.add_argument("--timeout", type=int, default=900)
    ap.add_argument("--env", default="", help="extra KEY=VALUE,... for every worker (e.g. a safe-sync knob under test)")
    # worker mode
    ap.add_argument("--worker", action="store_true")
    ap.add_argument("--model", default="SH1")
    ap.add_argument("--proc", type=int, default=0)
    args = ap.parse_args()
    if args.worker:
        return worker(args)
    os.makedirs(args.out, exist_ok=True)
    runs = []
    t0 = time.time()
    base_env = dict(os.environ)
    for kv in filter(None, args.env.split(",")):
        k, v = kv.split("="); base_env[k] = v
    for model in args.models.split(","):
        for p in range(args.procs):
            cmd = [sys.executable, os.path.abspath(__file__), "--worker", "--model", model, "--proc", str(p), "--n", str(args.n), "--frames", str(args.frames),
                   "--width", str(args.width), "--height", str(args.height), "--reps", str(args.reps), "--iters", str(args.iters), "--modes", args.modes, "--variants", args.variants]
            try:
                out = subprocess.run(cmd, capture_output=True, text=True, timeout=args.timeout, env=base_env)
                got = [json.loads(l[5:]) for l in out.stdout.splitlines() if l.startswith("SOAK ")]
                if out.returncode != 0:
                    got.append(dict(model=model, mode="process", variant="-", proc=p, rep=-1, cp=[], labels=[], err=f"rc {out.returncode}: {out.stderr[-300:]}"))
            except subprocess.TimeoutExpired:
                got = [dict(model=model, mode="process", variant="-", proc=p, rep=-1, cp=[], labels=[], err="timeout")]
            runs += got
        print(f"[soak] {model}: {len(runs)} runs so far, {time.time() - t0:.0f} s", flush=True)
    report, bad = analyse(runs)
    tag = f"N{args.n}_F{args.frames}"
    with open(os.path.join(args.out, f"soak_{tag}_runs.jsonl"), "w") as f:
        for r in runs:
            f.write(json.dumps(dict(r, labels=None)) + "\n")
    summary = dict(scene=tag, procs=args.procs, reps=args.reps, variants=args.variants.split(","), extra_env=args.env, total_runs=len(runs), deviating_runs=bad, wall_s=round(time.time() - t0, 1), groups=report)
    with open(os.path.join(args.out, f"soak_{tag}_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps(summary, indent=1))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main() or 0)
