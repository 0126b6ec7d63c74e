#!/usr/bin/env python3
"""Static attribution of a kernel's gfx950 instructions to source lines.

    hipcc --offload-arch=gfx950 -O3 -gline-tables-only -S --cuda-device-only -o k.s one_kernel.hip
    isa_lines.py k.s [kernel-name-substring] [--top N] [--loop]

--loop: only the instructions of the kernel's LARGEST loop (the basic blocks LLVM annotates "in Loop: Header=<label>", the
header and every nested loop included): for the wave-autonomous kernels that is the per-block loop, so the totals are what
one block executes when every branch is taken.

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


def loop_blocks(lines, want):
    """labels of the basic blocks of the largest depth-1 loop of the first matching kernel (nested loops included)"""
    inside = False
    size = collections.Counter()
    member = collections.defaultdict(set)
    parent = {}
    label, hdr = None, None
    for line in lines:
        m = re.match(r"^(_Z\w+):", line)
        if m:
            inside = want is None or want in m.group(1)
            continue
        if not inside:
            continue
        if line.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):(.*)$", line)
        if m:
            label = m.group(1)
            hdr = None
            h = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", m.group(2))
            if h:
                hdr = ".L" + h.group(1)
            continue
        if label is None:
            continue
        if "Loop Header: Depth=" in line:  # "=>This Loop Header" / "Parent Loop BBx_y Depth=1" comment lines under a label
            hdr = label
            continue
        pm = re.search(r"Parent Loop (BB\d+_\d+) Depth=", line)
        if pm:
            parent[label] = ".L" + pm.group(1)
            continue
        if re.match(r"\s+[a-z]\w+", line) and hdr:
            member[hdr].add(label)
            size[hdr] += 1
    # fold inner loops into their parents
    for inner, outer in parent.items():
        member[outer] |= member.get(inner, set()) | {inner}
        size[outer] += size.get(inner, 0)
    if not size:
        return set()
    top = max(size, key=size.get)
    return member[top] | {top}


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
    lines = open(path).read().splitlines()
    keep = None
    if "--loop" in sys.argv:
        keep = loop_blocks(lines, want)
    label = None
    for line in lines:
        mlab = re.match(r"^(\.LBB\d+_\d+):", line)
        if mlab:
            label = mlab.group(1)
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
        if m and cur and (keep is None or label in keep):
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
