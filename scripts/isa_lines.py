#!/usr/bin/env python3
"""Static attribution of a kernel's gfx950 instructions to source lines.

    hipcc --offload-arch=gfx950 -O3 -gline-tables-only -S --cuda-device-only -o k.s one_kernel.hip
    isa_lines.py k.s [kernel-name-substring] [--top N] [--by-func]

Counts VALU / SALU / LDS / VMEM instructions per `.loc file line` of the (first matching) kernel.  Static counts: a line
inside the unrolled per-slot loops appears once per slot, code under a skipped branch still counts.
"""
import collections
import re
import sys


def classify(op):
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "xlane"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return None


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    path = args[0]
    want = args[1] if len(args) > 1 else None
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 60
    files = {}
    per = collections.defaultdict(lambda: collections.Counter())
    cur = None
    inside = False
    tot = collections.Counter()
    for line in open(path):
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
            continue
        m = re.match(r"^(_Z\w+):", line)
        if m:
            inside = want is None or want in m.group(1)
            continue
        if not inside:
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            inside = False
            continue
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
        if m:
            cur = (int(m.group(1)), int(m.group(2)))
            continue
        m = re.match(r"\s+([a-z]\w+)", line)
        if m and cur:
            c = classify(m.group(1))
            if c:
                per[cur][c] += 1
                tot[c] += 1
    print("total", dict(tot))
    rows = sorted(per.items(), key=lambda kv: -(kv[1]["valu"] + kv[1]["xlane"]))
    for (f, l), c in rows[:top]:
        print(f"{files.get(f, f)}:{l:<5d} valu {c['valu']:4d} xlane {c['xlane']:3d} salu {c['salu']:4d} lds {c['lds']:4d} vmem {c['vmem']:3d}")


if __name__ == "__main__":
    main()
