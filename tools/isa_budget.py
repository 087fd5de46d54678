#!/usr/bin/env python3
"""Instruction budget of a sweep's per-observation loop, from the compiler's own ISA.

    python tools/isa_budget.py dist k_sweep_distILi0ELi1ELi0E      # file stem under psgradientsdf_amd/csrc, mangled-name fragment of the instance

Compiles the file for gfx950 with -gline-tables-only -S (same flags as the Makefile otherwise), takes the named kernel, finds its hot loop (the
loop with the most instructions among those that load image taps) and counts the instructions of the loop body
  * by CLASS  -- fp32 multiply-add family, other fp32 VALU, conversions, transcendental (quarter rate), integer / address VALU, compares and
                 selects, moves, vector memory, LDS, scalar ALU, scalar memory, waits / branches;
  * by STAGE  -- the source function the instruction was inlined from (device_common.h: project, sample, rendered, robust weights ...) or, for
                 code written in the kernel itself, the statement of the kernel it belongs to (matched by keywords).
Static counts of the loop body: blocks that only edge cases take (nearest-pixel sampling at the image border, NaN guards) are included and listed
separately as `cold` when they sit behind a forward branch that skips them (size given).  Prints a markdown table."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "psgradientsdf_amd", "csrc")


def classify(op):
    if op.startswith(("v_fma", "v_fmac", "v_mac", "v_mad_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_pk_fma", "v_pk_mul", "v_pk_add")):
        return "fp32 mul/add/fma"
    if op.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64")):
        return "fp64"
    if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_log", "v_exp", "v_sin", "v_cos")):
        return "transcendental (1/4 rate)"
    if op.startswith("v_cvt"):
        return "conversion"
    if op.startswith(("v_cmp", "v_cndmask", "v_max", "v_min", "v_med3", "v_cmpx")):
        return "compare / select / min-max"
    if op.startswith(("v_mov", "v_readlane", "v_readfirstlane", "v_writelane", "v_accvgpr", "v_swap")):
        return "move"
    if op.startswith(("v_mad_u32", "v_mad_i32", "v_mad_u64", "v_mad_i64", "v_mul_lo", "v_mul_hi", "v_mul_u32", "v_mul_i32", "v_add_u32", "v_add_co", "v_addc", "v_sub_u32", "v_sub_co", "v_subb",
                      "v_lshl", "v_lshr", "v_ashr", "v_and", "v_or", "v_xor", "v_not", "v_bfe", "v_bfi", "v_add3", "v_lshl_add", "v_add_lshl", "v_ffb", "v_bcnt", "v_mbcnt", "v_perm", "v_alignbit", "v_subrev_u32", "v_subrev_co", "v_mul_u", "v_add_i32")):
        return "integer / address VALU"
    if op.startswith(("v_frexp", "v_ldexp", "v_fract", "v_floor", "v_trunc", "v_rndne", "v_ceil", "v_div_", "v_mul_legacy", "v_fmaak", "v_fmamk")):
        return "other fp32 VALU"
    if op.startswith("v_"):
        return "other VALU"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vector memory"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("s_load", "s_buffer_load")):
        return "scalar memory"
    if op.startswith(("s_waitcnt", "s_nop", "s_branch", "s_cbranch", "s_barrier", "s_sleep", "s_setprio", "s_endpgm")):
        return "wait / branch"
    if op.startswith("s_"):
        return "scalar ALU"
    return "other"


def func_table(path):
    """line -> enclosing function name of a source file (by its `__device__` / `__global__` definitions)"""
    names, cur = {}, None
    pat = re.compile(r"^\s*(?:template\s*<[^>]*>\s*)?(?:__host__\s+)?(?:__device__|__global__)[^;{]*?\b([A-Za-z_][A-Za-z0-9_]*)\s*\(")
    lines = open(path).read().split("\n")
    pending_template = False
    for i, ln in enumerate(lines, 1):
        m = pat.match(ln)
        if m and "(" in ln:
            cur = m.group(1)
        names[i] = cur
    return names, lines


KERNEL_STAGES = [      # (keyword in the kernel's own source line, stage)
    ("FOR_EACH_VISIBLE_FRAME", "frame iteration (next set bit, record address)"), ("frame_at", "frame iteration (next set bit, record address)"),
    ("project_jac", "projection (Jacobian's own)"), ("project(", "projection"), ("sample<", "sample: taps, bilinear, image gradient"), ("rendered<", "shading (rendered intensity)"),
    ("pi_rows", "pi_grad rows x R^T"), ("dot3(U, dx", "Jacobian: image term"), ("gu[ch] * sq", "Jacobian: image term"), ("fp.l[1] * dn", "Jacobian: shading term"), ("lw[0]", "Jacobian: shading term"), ("v.rho[ch] * sq", "Jacobian: shading term"),
    ("robust_weight", "robust weight / loss"), ("robust_loss", "robust weight / loss"), ("pj.ok ? w", "robust weight / loss"),
    ("J[p][ch] * w", "accumulate 10 + 4"), ("B[q++]", "accumulate 10 + 4"), ("g[p] +=", "accumulate 10 + 4"), ("Ef += l", "accumulate 10 + 4"),
    ("mul3(fp.R, pr.p", "LED fall-off terms"), ("div_by", "LED fall-off terms"), ("dm", "LED fall-off terms"),
    ("acc", "accumulate"), ("H[", "accumulate"), ("rhs", "accumulate"),
]
FUNC_STAGES = {"project": "projection", "project_jac": "projection (Jacobian's own)", "sample": "sample: taps, bilinear, image gradient", "sample_cell": "sample: taps, bilinear, image gradient", "tap": "sample: taps, bilinear, image gradient",
               "rendered": "shading (rendered intensity)", "SH": "shading (rendered intensity)", "pi_rows": "pi_grad rows x R^T", "pi_rows_world": "pi_grad rows x R^T", "robust_weight": "robust weight / loss", "robust_loss": "robust weight / loss",
               "frame_at": "frame iteration (next set bit, record address)", "next_frame": "frame iteration (next set bit, record address)", "dot3": None, "mul3": None, "mulT3": None, "norm3": None, "div_by": "LED fall-off terms"}


def main():
    stem, frag = sys.argv[1], sys.argv[2]
    src = os.path.join(CSRC, stem + ".hip")
    out = f"/tmp/isa_budget_{stem}.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-slp-vectorize", "-gline-tables-only", "-S", "--cuda-device-only", "-c", src, "-o", out],
                          cwd=CSRC, stderr=subprocess.DEVNULL)
    text = open(out).read().split("\n")
    start = next(i for i, l in enumerate(text) if l.startswith("_Z") and ":" in l and frag in l.split(":")[0])
    end = next(i for i in range(start, len(text)) if text[i].strip().startswith("s_endpgm"))
    body = text[start:end + 1]
    kfuncs, klines = func_table(src)
    hfuncs, hlines = func_table(os.path.join(CSRC, "device_common.h"))
    hlines_pre = hlines
    # instructions with their location chain
    insts = []      # (index in body, opcode, innermost (file, line), kernel-file line or None, text, cold?)
    cur_inner, cur_kline, cur_cold = None, None, False
    # cold code by construction: the image-border sampler and the "Jacobian's projection fell into the neighbouring pixel cell" redo (~1e-5 of the observations)
    redo_lines = {i for i, l in enumerate(hlines_pre, 1) if "sample_cell<true>(base, frame" in l or "sample_u8_cell<true>(base, scale, frame" in l}
    labels = {}
    for i, l in enumerate(body):
        s = l.strip()
        m = re.match(r"\.loc\s+\d+\s+\d+\s+\d+.*?;\s*(\S+?):(\d+):\d+(.*)$", s)
        if m:
            cur_inner = (os.path.basename(m.group(1)), int(m.group(2)))
            chain = re.findall(r"([A-Za-z0-9_./-]+):(\d+):\d+", m.group(3))
            cur_kline = None
            if cur_inner[0] == os.path.basename(src):
                cur_kline = cur_inner[1]
            for f, ln in chain:
                if os.path.basename(f) == os.path.basename(src):
                    cur_kline = int(ln)
            allloc = [cur_inner] + [(os.path.basename(f), int(ln)) for f, ln in chain]
            cur_cold = any(f == "device_common.h" and (hfuncs.get(ln) == "sample_border" or hfuncs.get(ln) == "pix" or ln in redo_lines) for f, ln in allloc)
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if not s or s.startswith((";", ".", "_Z")) or s.endswith(":"):
            continue
        op = s.split()[0]
        insts.append((i, op, cur_inner, cur_kline, s, cur_cold))
    # loops: backward branches
    loops = collections.defaultdict(list)
    for n, (i, op, inner, kl, s, cold) in enumerate(insts):
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = s.split()[-1]
            if tgt in labels and labels[tgt] <= n:
                loops[tgt].append(n)
    best = None
    for tgt, ends in loops.items():
        a, b = labels[tgt], max(ends)
        seg = insts[a:b + 1]
        nload = sum(1 for x in seg if x[1].startswith(("global_load", "buffer_load")) and x[2] and x[2][0] == "device_common.h" and (hfuncs.get(x[2][1]) or "").startswith(("sample", "tap")))
        if nload and (best is None or (b - a) > (best[1] - best[0])):
            best = (a, b, tgt)
    if best is None:
        print("no loop with image taps found"); return
    a, b, tgt = best
    seg = insts[a:b + 1]

    def stage_of(x):
        _, op, inner, kl, s, cold = x
        if cold:
            return "COLD: image border / neighbouring-cell redo (~1e-5 of the observations)"
        if inner and inner[0] == "device_common.h":
            fn = hfuncs.get(inner[1])
            st = FUNC_STAGES.get(fn, "?") if fn in FUNC_STAGES else None
            if st:
                return st
        if kl and 0 < kl <= len(klines):
            srcl = klines[kl - 1]
            for key, st in KERNEL_STAGES:
                if key in srcl:
                    return st
            return "kernel line: " + srcl.strip()[:60]
        if inner and inner[0] == "device_common.h":
            return "device_common.h: " + str(hfuncs.get(inner[1]))
        return "unattributed"

    by_class = collections.Counter(classify(x[1]) for x in seg)
    by_stage = collections.defaultdict(collections.Counter)
    for x in seg:
        by_stage[stage_of(x)][classify(x[1])] += 1
    valu = lambda c: sum(v for k, v in c.items() if k in ("fp32 mul/add/fma", "fp64", "transcendental (1/4 rate)", "conversion", "compare / select / min-max", "move", "integer / address VALU", "other fp32 VALU", "other VALU"))
    hot = [x for x in seg if not x[5]]
    by_class_hot = collections.Counter(classify(x[1]) for x in hot)
    print(f"### `{frag}` ({stem}.hip): hot loop `{tgt}`, {len(seg)} instructions in the loop body ({valu(by_class)} VALU); on the path every observation takes: {len(hot)} ({valu(by_class_hot)} VALU)\n")
    by_class = by_class_hot
    print("| class | instructions |\n|---|---|")
    for k, v in by_class.most_common():
        print(f"| {k} | {v} |")
    print("\n| stage | VALU | fp32 mul/add/fma | integer / address | conversion | compare / select | move | transcendental | vector memory | LDS | scalar + wait |\n|---|---|---|---|---|---|---|---|---|---|---|")
    for st, c in sorted(by_stage.items(), key=lambda kv: -valu(kv[1])):
        print(f"| {st} | {valu(c)} | {c['fp32 mul/add/fma']} | {c['integer / address VALU']} | {c['conversion']} | {c['compare / select / min-max']} | {c['move']} | {c['transcendental (1/4 rate)']} | {c['vector memory']} | {c['LDS']} | {c['scalar ALU'] + c['scalar memory'] + c['wait / branch']} |")


if __name__ == "__main__":
    main()
