#!/usr/bin/env python3
"""profiles/pmc_summary.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py: HBM-side bytes per launch
of every bench kernel, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: both counters are in
KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so it is doubled; WRITE_SIZE is taken as is (uncalibrated).

    tools/pmc_summary.py <fetch pmc_counter_collection.csv> <write pmc_counter_collection.csv> > profiles/pmc_summary.json
"""
import csv, json, re, sys
from collections import defaultdict

NAMES = {"k_cgf_solve": "pcg_solve", "k_cgp_solve": "pcg_solve", "k_cgf_pass": "pcg_pass", "k_cgf_init": "pcg_init", "k_sweep_dist": "sweep_dist", "k_sweep_pose": "sweep_pose",
         "k_sweep_light": "sweep_light", "k_sweep_albedo": "sweep_albedo", "k_energy": "energy", "k_assemble": "assemble",
         "k_derive": "derive", "k_apply_albedo": "apply_albedo", "k_apply_dist": "apply_dist"}


def load(path, counter):
    acc = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            m = re.search(r"psg::(k_[a-z0-9_]+)", row["Kernel_Name"])
            if m and m.group(1) in NAMES:
                acc[NAMES[m.group(1)]].append(float(row["Counter_Value"]))
    return acc


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    if not fetch[k] and not write[k]:
        continue
    f = sum(fetch[k]) / max(len(fetch[k]), 1); w = sum(write[k]) / max(len(write[k]), 1)
    out[k] = {"launches": len(fetch[k]), "fetch_size_kib_raw": f, "write_size_kib_raw": w,
              "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
              "note": "FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, KiB -> bytes; Infinity-Cache hits are counted"}
import os
out["commit"] = os.environ.get("PSGSDF_COMMIT", "?")
try:
    out["source_hash"] = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "psgradientsdf_amd", "csrc", ".build_hash")).read().strip()[:12]
except OSError:
    out["source_hash"] = None
json.dump(out, sys.stdout, indent=1)
print()
