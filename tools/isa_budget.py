#!/usr/bin/env python3
"""Static instruction budget of one kernel from hipcc's assembly (-S -gline-tables-only).

    hipcc --offload-arch=gfx950 <product flags> -gline-tables-only --cuda-device-only -S rssync_kernels.hip -o k.s
    python tools/isa_budget.py k.s 'lmeds_kernelILi8ELi0ELi80' [--blocks]

Splits the kernel into basic blocks, classifies every instruction (VALU / packed VALU / SALU / LDS / VMEM / SMEM /
branch / wait-barrier) and attributes it to a STAGE by the source line of its innermost .loc (file + line ranges
below).  --blocks prints the per-block view used to assign trip counts.
"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_pk_"):
        return "valu_pk"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith(("s_waitcnt", "s_barrier", "s_nop", "s_sleep", "s_endpgm", "s_setprio")):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def parse(path, pattern):
    lines = open(path).read().split("\n")
    files = {}
    start = None
    for i, l in enumerate(lines):
        m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
        if m:
            files[int(m.group(1))] = m.group(2)
        if start is None and re.match(r"^_Z\w*%s\w*:" % pattern, l):
            start = i
    if start is None:
        raise SystemExit("kernel not found")
    blocks = []
    cur = {"label": "entry", "ins": []}
    loc = ("?", 0)
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur)
            cur = {"label": m.group(1), "ins": []}
            continue
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            loc = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        m = re.match(r"^\s+([a-z_0-9]+)(\s|$)", l)
        if m and not l.strip().startswith("."):
            op = m.group(1)
            tgt = None
            if op.startswith(("s_cbranch", "s_branch")):
                t = re.search(r"(\.LBB\d+_\d+)", l)
                tgt = t.group(1) if t else None
            cur["ins"].append((op, classify(op), loc, tgt))
    blocks.append(cur)
    return blocks


def main():
    path, pattern = sys.argv[1], sys.argv[2]
    blocks = parse(path, pattern)
    if "--blocks" in sys.argv:
        for b in blocks:
            c = collections.Counter(k for _, k, _, _ in b["ins"])
            locs = collections.Counter("%s:%d" % l for _, _, l, _ in b["ins"])
            tg = [t for _, _, _, t in b["ins"] if t]
            top = ", ".join("%s x%d" % kv for kv in locs.most_common(4))
            print("%-12s n=%4d  valu %4d pk %4d salu %4d lds %3d vmem %3d br %d -> %s | %s" % (
                b["label"], len(b["ins"]), c["valu"], c["valu_pk"], c["salu"], c["lds"], c["vmem"], c["branch"],
                ",".join(tg), top))
    tot = collections.Counter()
    for b in blocks:
        for _, k, _, _ in b["ins"]:
            tot[k] += 1
    print("static totals:", dict(tot))


if __name__ == "__main__":
    main()
