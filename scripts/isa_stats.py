#!/usr/bin/env python3
"""Occupancy / spill guard for the gfx950 code objects of libasciichat_hip.so.

Extracts the device code objects from the built library, reads every kernel's AMDGPU metadata note (.vgpr_count,
.sgpr_count, spill counts, LDS) and fails when a kernel leaves the register budget its launch geometry depends on:

  * render_frames_kernel, 512- and 256-thread geometries, non-half-block modes: <= 128 VGPRs (two / four workgroups
    per CU; at 130 only one 512-thread workgroup fits and the step time doubles) -- they are pinned there with
    amdgpu_waves_per_eu(4), which the compiler pays for with a handful of spilled VGPRs: at most 24, else the pin
    has stopped being cheap;
  * render_stream_kernel: <= 64 VGPRs (two 1024-thread or four 512-thread workgroups per CU; the truecolor-background
    mode, whose 48-byte tokens fill the LDS first, <= 72) and no scratch memory (SGPRs parked in VGPR lanes are fine);
  * every other kernel: no VGPR spills.

Usage: isa_stats.py [path/to/libasciichat_hip.so] [--out profiles/isa_stats.txt]   (exit status 1 on a violation)
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
HALFBLOCK = {5, 6, 7, 8}


def kernels_of(lib):
    tmp = tempfile.mkdtemp(prefix="isa_stats_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        out = []
        objs = [os.path.join(tmp, f) for f in sorted(os.listdir(tmp)) if "gfx950" in f]
        if objs and open(objs[0], "rb").read(4) != b"\x7fELF":
            # --offload-compress builds (the product's): the fat binary is a sequence of compressed bundles ("CCOB" + a
            # header that carries the bundle's size) which llvm-objdump does not split; clang-offload-bundler unpacks one
            objs = []
            fat = os.path.join(tmp, "fat.bin")
            subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", local, fat], check=True)
            data = open(fat, "rb").read()
            at, k = data.find(b"CCOB"), 0
            while at >= 0:
                version = int.from_bytes(data[at + 4:at + 6], "little")
                size = int.from_bytes(data[at + 8:at + (16 if version >= 3 else 12)], "little") if version >= 2 else 0
                nxt = data.find(b"CCOB", at + max(size, 4))
                chunk = os.path.join(tmp, f"bundle{k}.bin")
                open(chunk, "wb").write(data[at:at + size] if size else data[at:nxt if nxt >= 0 else len(data)])
                co = os.path.join(tmp, f"bundle{k}.gfx950.co")
                subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + chunk,
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True,
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                objs.append(co)
                at, k = nxt, k + 1
        for obj in objs:
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", obj], check=True,
                                   capture_output=True, text=True).stdout
            cur = {}
            for line in notes.splitlines():
                m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
                if not m:
                    continue
                k, v = m.group(1), m.group(2).strip().strip("'")
                if k == "agpr_count" and cur.get("name"):  # first key of a kernel's map in the note
                    out.append(cur)
                    cur = {}
                cur[k] = v
            if cur.get("name"):
                out.append(cur)
        return [k for k in out if "vgpr_count" in k and k.get("name")]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def demangle(name):
    try:
        tool = os.path.join(LLVM, "llvm-cxxfilt")
        return subprocess.run([tool if os.path.exists(tool) else "c++filt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        return name


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else os.path.join(ROOT, "ascii-chat_amd", "libasciichat_hip.so")
    out_path = None
    if "--out" in sys.argv:
        out_path = sys.argv[sys.argv.index("--out") + 1]
        if out_path in args:
            args.remove(out_path)
            lib = args[0] if args else os.path.join(ROOT, "ascii-chat_amd", "libasciichat_hip.so")
    rows, bad = [], []
    for k in kernels_of(lib):
        name = k["name"]
        vg, sg = int(k["vgpr_count"]), int(k["sgpr_count"])
        vs, ss = int(k.get("vgpr_spill_count", 0)), int(k.get("sgpr_spill_count", 0))
        lds = int(k.get("group_segment_fixed_size", 0))
        scratch = int(k.get("private_segment_fixed_size", 0))
        limit, why = None, ""
        m = re.search(r"render_frames_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb([01])ELb([01])E", name)
        ms = re.search(r"render_stream_kernelILi(\d+)ELi(\d+)ELi(\d+)ELb([01])ELb([01])ELi(\d+)ELb([01])ELb([01])E", name)
        mr = re.search(r"render_rows_kernelILi(\d+)ELi(\d+)ELi(\d+)ELb([01])ELb([01])ELb([01])ELb([01])E", name)
        if mr:
            mode, waves, cpl = int(mr.group(1)), int(mr.group(2)), int(mr.group(3))
            crc = mr.group(5) == "1"
            short = (f"render_rows_kernel<mode {mode}, {waves} waves, {cpl} cells/lane, generic {mr.group(4)}"
                     f"{', +crc' if crc else ''}{', rows cut into segments' if mr.group(6) == '1' else ''}"
                     f"{', frames shared out' if mr.group(7) == '1' else ''}>")
            if not crc and not (mr.group(4) == "1" and mode == 7 and cpl > 4):
                limit, why = 128, "4 waves per SIMD: two 8-wave workgroups per CU"
        elif m:
            mode, block = int(m.group(1)), int(m.group(2))
            short = f"render_frames_kernel<mode {mode}, {block} thr, cap {m.group(3)}, comp {m.group(5)}, split {m.group(6)}>"
            if block in (256, 512) and int(m.group(3)) <= 2048 and mode not in HALFBLOCK:
                limit, why = 128, "two (four) workgroups per CU"
        elif ms:
            mode, waves, cpl = int(ms.group(1)), int(ms.group(2)), int(ms.group(3))
            crc, pack = ms.group(5) == "1", int(ms.group(6))
            short = (f"render_stream_kernel<mode {mode}, {waves} waves, {cpl} cells/lane, generic {ms.group(4)}"
                     f"{', +crc' if crc else ''}{', exact-length frames' if pack else ''}{' + frame crc' if pack == 2 else ''}"
                     f"{', frames shared out' if ms.group(7) == '1' else ''}{', length-first' if ms.group(8) == '1' else ''}>")
            if crc or pack:  # the checksum's 16-byte groups are in flight next to the following block's samples; the
                             # exact-length form stages a whole frame in LDS: one 16-wave workgroup per CU either way
                limit, why = 128, "4 waves per SIMD: one 16-wave or two 8-wave workgroups per CU (LDS allows no more)"
            elif ms.group(7) == "1":  # shared-out frames are small launches: a few four-wave workgroups per CU, all resident
                limit, why = 72, "7 waves per SIMD (small launches: every workgroup is resident anyway)"
            elif ms.group(8) == "1":  # exact-length frames, the loop run twice: its own instantiation so that the plain one keeps 8
                limit, why = 72, "7 waves per SIMD"
            else:
                # (mode 17 = truecolor foreground with multi-byte glyphs, ACHIP_STREAM_MODE_TRUE_FG_U8: its RLE-state chain next
                # to the tokens; 4 = truecolor background)
                limit, why = (72, "7 waves per SIMD") if mode in (4, 17) else (64, "8 waves per SIMD")
        else:
            short = demangle(name).split("(")[0].replace("void ", "")[-70:]
        problems = []
        if limit is not None and vg > limit:
            problems.append(f"{vg} VGPRs > {limit} ({why})")
        if vs > 0:
            problems.append(f"{vs} VGPRs spilled")
        if scratch > 0:  # no product kernel touches scratch memory
            problems.append(f"{scratch} B scratch")
        rows.append((short, vg, sg, vs, ss, scratch, lds, "; ".join(problems)))
        if problems:
            bad.append((short, problems))
    rows.sort()
    lines = [f"# gfx950 kernel resources of {os.path.relpath(lib, ROOT)} (scripts/isa_stats.py; llvm-readelf --notes)",
             f"# {'kernel':92s} {'VGPR':>5s} {'SGPR':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'LDS':>7s}"]
    for r in rows:
        lines.append(f"  {r[0]:92s} {r[1]:5d} {r[2]:5d} {r[3]:6d} {r[4]:6d} {r[5]:7d} {r[6]:7d}  {r[7]}")
    lines.append(f"# {len(rows)} kernels, {len(bad)} violation(s)")
    text = "\n".join(lines) + "\n"
    if out_path:
        with open(out_path, "w") as f:
            f.write(text)
    else:
        sys.stdout.write(text)
    for short, problems in bad:
        print(f"isa_stats: {short}: {', '.join(problems)}", file=sys.stderr)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
