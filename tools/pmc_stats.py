#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (pmc_counter_collection.csv) per kernel: mean counter value per dispatch."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"psg::(k_[a-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:50]


def load(path):
    acc = defaultdict(lambda: defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


def main(paths):
    merged = defaultdict(dict)
    for p in paths:
        for k, cs in load(p).items():
            for c, v in cs.items():
                merged[k][c] = (sum(v) / len(v), len(v))
    counters = sorted({c for k in merged for c in merged[k]})
    print("| kernel | n | " + " | ".join(counters) + " |")
    print("|---|---|" + "---|" * len(counters))
    for k in sorted(merged, key=lambda k: -merged[k].get("SQ_WAVE_CYCLES", merged[k].get("FETCH_SIZE", (0, 0)))[0]):
        n = max(v[1] for v in merged[k].values())
        print(f"| {k} | {n} | " + " | ".join(f"{merged[k][c][0]:.4g}" if c in merged[k] else "" for c in counters) + " |")


if __name__ == "__main__":
    main(sys.argv[1:])
