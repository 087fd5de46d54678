#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel stats table (markdown)."""
import re
import sqlite3
import sys


def short(name):
    m = re.search(r"psg::(k_[a-z0-9_]+)(<[^>]*>)?", name)
    if m:
        return m.group(1) + (m.group(2) or "")
    return name[:60]


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        k = short(name)
        a = agg.setdefault(k, [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    lines = ["| kernel | calls | total us | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"| {k} | {a[0]} | {a[1]:.1f} | {a[1]/a[0]:.2f} | {a[2]:.2f} | {a[3]:.2f} | {100*a[1]/tot:.1f} |")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")
    return agg


def per_iteration(agg, json_out, commit):
    """us per Gauss-Newton iteration of every kernel of the product loop (bench.py `kernels_rocprof`): total time / launches of the solve kernel,
    which runs exactly once per iteration"""
    import json
    solves = sum(a[0] for k, a in agg.items() if k.startswith(("k_cgp_solve", "k_cgf_solve")))
    if not solves:
        return
    loop = ("k_cgp_solve", "k_cgf_solve", "k_sweep_", "k_derive", "k_solve_", "k_energy", "k_apply_", "k_sum_parts", "k_restore_", "k_frame_cols", "k_marker")
    # a kernel of the loop runs once per iteration (the traced runs also hold the loops' openings and closings: a few extra sweeps, energy evaluations
    # and undo kernels per psgsdf_optimize call, listed with their launch counts)
    per = {k: {"avg_us": round(a[1] / a[0], 2), "launches": a[0]} for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]) if k.startswith(loop)}
    once = sum(v["avg_us"] for k, v in per.items() if v["launches"] >= solves)
    import os
    try:
        src_hash = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "psgradientsdf_amd", "csrc", ".build_hash")).read().strip()[:12]
    except OSError:
        src_hash = None
    json.dump({"source": "rocprofv3 --kernel-trace of `python bench.py --no-cpu-baseline --no-extra` (tools/profile_round.sh)", "commit": commit, "source_hash": src_hash, "iterations_traced": solves,
               "kernels": per, "us_per_iteration": round(once, 1), "note": "us_per_iteration = sum of the average durations of the kernels that run once per iteration (launches >= iterations_traced)"}, open(json_out, "w"), indent=1)


if __name__ == "__main__":
    import os
    agg = main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
    if len(sys.argv) > 3:
        per_iteration(agg, sys.argv[3], os.environ.get("PSGSDF_COMMIT", "?"))
