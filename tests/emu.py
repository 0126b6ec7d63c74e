"""Builds and drives the CPU fiber emulation of the HIP kernels (tests/hipemu). TESTS ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

from achip_ctypes import Composite, Frame, Lut, bind_host  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ascii-chat_amd", "csrc")
INC = os.path.join(ROOT, "include")
EMU_DIR = os.path.join(ROOT, "tests", "hipemu")
EMU_SO = os.path.join(EMU_DIR, "_build", "libachip_emu.so")

_lib = None


def build():
    srcs = [os.path.join(EMU_DIR, "emu_driver.cpp"), os.path.join(EMU_DIR, "hip_emu.h"),
            os.path.join(CSRC, "render_kernels.hpp"), os.path.join(CSRC, "render_stream.hpp"), os.path.join(CSRC, "render_rows.hpp"), os.path.join(CSRC, "stream_kernels.hpp"), os.path.join(CSRC, "crc_kernels.hpp"), os.path.join(CSRC, "crc_math.hpp"), os.path.join(EMU_DIR, "gfx950_ops.hpp"),
            os.path.join(CSRC, "render_variants.h"),
            os.path.join(INC, "achip_types.h"), os.path.join(CSRC, "achip_host.c"), os.path.join(INC, "achip_host.h")]
    if os.path.exists(EMU_SO) and all(os.path.getmtime(s) <= os.path.getmtime(EMU_SO) for s in srcs):
        return EMU_SO
    os.makedirs(os.path.dirname(EMU_SO), exist_ok=True)
    obj = os.path.join(os.path.dirname(EMU_SO), "achip_host.%d.o" % os.getpid())
    tmp = EMU_SO + ".%d.tmp" % os.getpid()
    subprocess.check_call(["gcc", "-std=gnu11", "-O2", "-fPIC", "-I" + INC, "-c", os.path.join(CSRC, "achip_host.c"), "-o", obj])
    extra = os.environ.get("ACHIP_EMU_DEFS", "").split()  # e.g. -DACHIP_EMIT_OR_MODES=0x3FF to test an experiment
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-DACHIP_ALL_GEOMETRIES", *extra, "-I" + EMU_DIR, "-I" + CSRC, "-I" + INC,  # EMU_DIR first: <gfx950_ops.hpp> = the emulator's twin
                           os.path.join(EMU_DIR, "emu_driver.cpp"), obj, "-o", tmp])
    os.replace(tmp, EMU_SO)  # atomic: a concurrent test process (pytest -n) never loads a half-written library
    return EMU_SO


def lib():
    global _lib
    if _lib is None:
        _lib = bind_host(C.CDLL(build()))
        _lib.emu_render_batch.restype = C.c_int
        _lib.emu_render_batch.argtypes = [C.c_int, C.c_int, C.POINTER(Frame), C.c_int, C.POINTER(Lut), C.c_void_p,
                                          C.c_uint64, C.c_void_p]
        _lib.emu_render_stream_crc.restype = C.c_int
        _lib.emu_render_stream_crc.argtypes = [C.c_int, C.c_int, C.POINTER(Frame), C.c_int, C.POINTER(Lut), C.c_void_p,
                                               C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.emu_render_rows_crc.restype = C.c_int
        _lib.emu_render_rows_crc.argtypes = _lib.emu_render_stream_crc.argtypes
        _lib.emu_rep_rule_check.restype = C.c_long
        _lib.emu_rep_rule_check.argtypes = []
        _lib.emu_quant16_check.restype = C.c_long
        _lib.emu_quant16_check.argtypes = [C.POINTER(C.c_uint32)]
        _lib.emu_set_uniform.restype = C.c_int
        _lib.emu_set_uniform.argtypes = [C.c_int]
        _lib.emu_set_parts.restype = None
        _lib.emu_set_parts.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_uint32]
        _lib.emu_resize_nn.restype = None
        _lib.emu_resize_nn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_uint32,
                                       C.c_uint32]
        _lib.emu_tint.restype = None
        _lib.emu_tint.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int]
        _lib.emu_flip.restype = None
        _lib.emu_flip.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_int]
        _lib.emu_crc32c.restype = None
        _lib.emu_crc32c.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.emu_crc_mulmod.restype = C.c_uint32
        _lib.emu_crc_mulmod.argtypes = [C.c_uint32, C.c_uint32]
        _lib.emu_crc_pow.restype = C.c_uint32
        _lib.emu_crc_pow.argtypes = [C.c_uint32, C.c_uint64]
        _lib.emu_crc_x8_pow2.restype = C.c_uint32
        _lib.emu_crc_x8_pow2.argtypes = [C.c_int]
        _lib.emu_pack.restype = None
        _lib.emu_pack.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p,
                                  C.c_void_p, C.c_int]
        _lib.emu_composite.restype = None
        _lib.emu_composite.argtypes = [C.POINTER(Composite), C.c_void_p]
    return _lib


def make_lut(palette):
    lut = Lut()
    p = palette.encode("utf-8") if isinstance(palette, str) else palette
    assert lib().achip_lut_build(p, C.byref(lut)) == 0
    return lut


_EPOCH = [0]


def render_frames(mode, frames, palette, variant=0, stride=None, rows_per_part=0, uniform=False, parts=0, sync=None,
                  line_phase=None):
    """uniform: pass the batch's common descriptor by value when it has one (what plans do by default).
    rows_per_part > 0 renders every frame with ceil(rows / rows_per_part) workgroups (multi-part frames).
    parts > 1 (stream geometries): a frame's blocks shared out over that many workgroups; sync (numpy uint64[n * parts])
    may be passed in to launch again on words that hold an earlier launch's epochs."""
    """frames: ctypes array/list of Frame (src pointers = host numpy memory). Returns list of bytes / int codes."""
    L = lib()
    n = len(frames)
    arr = (Frame * n)(*frames)
    lut = make_lut(palette)
    if stride is None:
        stride = max(int(L.achip_out_bound(mode, C.byref(arr[i]))) for i in range(n))
        stride = (stride + 1 + 15) // 16 * 16
    out = np.full(n * stride + 64 + 256, 0xEE, dtype=np.uint8)
    # 16-byte align the slab like hipMalloc would
    base = (out.ctypes.data + 15) // 16 * 16
    if line_phase is not None:  # the slab starts 16 * line_phase bytes behind a 128-byte line boundary (the drains'
        base = (out.ctypes.data + 127) // 128 * 128 + 16 * line_phase  # lane -> group mapping follows the ADDRESS)
    ln = np.zeros(n, dtype=np.uint32)
    if rows_per_part > 0:
        hb = mode in (5, 6, 7, 8)
        max_rows = max(((f.out_h + 1) // 2 if hb else f.out_h) for f in frames)
        bands = (max_rows + rows_per_part - 1) // rows_per_part
        band_sync = np.zeros(n * bands, dtype=np.uint64)
        _EPOCH[0] += 1
        L.emu_set_parts(bands, rows_per_part, band_sync.ctypes.data, _EPOCH[0])
    elif parts > 1:
        if sync is None:
            sync = np.zeros(n * parts, dtype=np.uint64)
        assert sync.size >= n * parts
        _EPOCH[0] += 1
        L.emu_set_parts(parts, 1, sync.ctypes.data, _EPOCH[0])
    L.emu_set_uniform(1 if uniform else 0)
    try:
        rc = L.emu_render_batch(mode, variant, arr, n, C.byref(lut), base, stride, ln.ctypes.data)
    finally:
        L.emu_set_parts(1, 0, None, 1)
        L.emu_set_uniform(0)
    assert rc == 0
    res = []
    for i in range(n):
        if ln[i] >= 0xFFFFFFF0:
            res.append(int(ln[i]))
        else:
            res.append(C.string_at(base + i * stride, int(ln[i])))
    if line_phase is not None:  # nothing outside the frames (and their NULs) was written
        o0 = base - out.ctypes.data
        assert (out[:o0] == 0xEE).all(), "bytes in front of the slab were written"
        for i in range(n):
            if ln[i] < 0xFFFFFFF0:
                gap = out[o0 + i * stride + int(ln[i]) + 1:o0 + (i + 1) * stride]
                assert (gap == 0xEE).all(), f"frame {i}: bytes behind the frame's NUL were written"
        assert (out[o0 + n * stride:] == 0xEE).all(), "bytes behind the slab were written"
    return res


def render_frames_crc(mode, frames, palette, variant=20, stride=None, dims=None):
    """The stream kernel with the frame CRC riding its drain: returns ([bytes | code], [crc]); with dims ([(w, h)] per
    frame) also the 24-byte packet headers and the CRCs of header || frame: ([..], [crc], [hdr bytes], [pkt crc])."""
    L = lib()
    n = len(frames)
    arr = (Frame * n)(*frames)
    lut = make_lut(palette)
    if stride is None:
        stride = max(int(L.achip_out_bound(mode, C.byref(arr[i]))) for i in range(n))
        stride = (stride + 1 + 15) // 16 * 16
    out = np.full(n * stride + 64, 0xEE, dtype=np.uint8)
    base = (out.ctypes.data + 15) // 16 * 16
    ln = np.zeros(n, dtype=np.uint32)
    crc = np.full(n, 0xDEADBEEF, dtype=np.uint32)
    d = np.array(dims, dtype=np.uint32).reshape(n, 2) if dims is not None else None
    hdr = np.full(n * 24, 0xEE, dtype=np.uint8)
    pkt = np.full(n, 0xDEADBEEF, dtype=np.uint32)
    entry = L.emu_render_rows_crc if variant >= 24 else L.emu_render_stream_crc
    assert entry(mode, variant, arr, n, C.byref(lut), base, stride, ln.ctypes.data, crc.ctypes.data,
                                   d.ctypes.data if d is not None else None, hdr.ctypes.data if d is not None else None,
                                   pkt.ctypes.data if d is not None else None) == 0
    res = [int(ln[i]) if ln[i] >= 0xFFFFFFF0 else C.string_at(base + i * stride, int(ln[i])) for i in range(n)]
    if d is None:
        return res, [int(c) for c in crc]
    return res, [int(c) for c in crc], [hdr[24 * i:24 * i + 24].tobytes() for i in range(n)], [int(c) for c in pkt]


def render_frames_packed(mode, frames, palette, variant=16, stride=None, dims=None, want_crc=True, capacity=None, cursor=None):
    """The stream kernel's PACK instantiations (frames at their exact lengths straight from the render, no slab): returns
    dict(lens=out_len[n], off=off_out[n + 1], plen=len_out[n], dst=bytes of the packed buffer, crc / hdr / pkt when want_crc,
    cursor=the two cursor words after the launch).  `cursor` may be passed in (a numpy uint64[2]) to launch again on it."""
    L = lib()
    L.emu_render_stream_pack.restype = C.c_int
    L.emu_render_stream_pack.argtypes = [C.c_int, C.c_int, C.POINTER(Frame), C.c_int, C.POINTER(Lut), C.c_uint64, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                         C.c_void_p, C.c_void_p]
    n = len(frames)
    arr = (Frame * n)(*frames)
    lut = make_lut(palette)
    if stride is None:
        stride = max(int(L.achip_out_bound(mode, C.byref(arr[i]))) for i in range(n))
        stride = (stride + 1 + 15) // 16 * 16
    cap = n * stride if capacity is None else capacity
    raw = np.full(n * stride + 64, 0xEE, dtype=np.uint8)
    base = (raw.ctypes.data + 15) // 16 * 16
    ln = np.zeros(n, dtype=np.uint32)
    off = np.zeros(n + 1, dtype=np.uint64)
    plen = np.zeros(n, dtype=np.uint32)
    cur = np.zeros(2, dtype=np.uint64) if cursor is None else cursor
    crc = np.zeros(n, dtype=np.uint32)
    hdr = np.zeros(24 * n, dtype=np.uint8)
    pkt = np.zeros(n, dtype=np.uint32)
    d = np.array(dims, dtype=np.uint32) if dims is not None else None
    rc = L.emu_render_stream_pack(mode, variant, arr, n, C.byref(lut), stride, ln.ctypes.data,
                                  crc.ctypes.data if want_crc else None, d.ctypes.data if d is not None else None,
                                  hdr.ctypes.data if (want_crc and d is not None) else None,
                                  pkt.ctypes.data if (want_crc and d is not None) else None, base, cap, off.ctypes.data,
                                  plen.ctypes.data, cur.ctypes.data)
    assert rc == 0
    view = np.ctypeslib.as_array((C.c_uint8 * (n * stride + 16)).from_address(base)).copy()
    return dict(lens=ln, off=off, plen=plen, dst=view, crc=crc, hdr=hdr, pkt=pkt, cursor=cur, stride=stride)


def render_frames_length_first(frames, palette, variant=17, stride=None, capacity=None, cursor=None, uniform=False):
    """The stream kernel's LENGTH-FIRST instantiation (exact-length truecolor-foreground frames of any size in one launch):
    dict(lens, off, plen, dst, cursor, stride) as render_frames_packed."""
    L = lib()
    L.emu_render_stream_lenfirst.restype = C.c_int
    L.emu_render_stream_lenfirst.argtypes = [C.c_int, C.POINTER(Frame), C.c_int, C.POINTER(Lut), C.c_uint64, C.c_void_p, C.c_void_p,
                                             C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    n = len(frames)
    arr = (Frame * n)(*frames)
    lut = make_lut(palette)
    if stride is None:
        stride = max(int(L.achip_out_bound(1, C.byref(arr[i]))) for i in range(n))
        stride = (stride + 1 + 15) // 16 * 16
    cap = n * stride if capacity is None else capacity
    raw = np.full(n * stride + 64 + 256, 0xEE, dtype=np.uint8)
    base = (raw.ctypes.data + 15) // 16 * 16
    ln = np.zeros(n, dtype=np.uint32)
    off = np.zeros(n + 1, dtype=np.uint64)
    plen = np.zeros(n, dtype=np.uint32)
    cur = np.zeros(2, dtype=np.uint64) if cursor is None else cursor
    L.emu_set_uniform(1 if uniform else 0)
    try:
        rc = L.emu_render_stream_lenfirst(variant, arr, n, C.byref(lut), stride, ln.ctypes.data, base, cap, off.ctypes.data,
                                          plen.ctypes.data, cur.ctypes.data)
    finally:
        L.emu_set_uniform(0)
    assert rc == 0
    view = np.ctypeslib.as_array((C.c_uint8 * (n * stride + 16)).from_address(base)).copy()
    return dict(lens=ln, off=off, plen=plen, dst=view, cursor=cur, stride=stride)


def frame_for_convert(img, width, height, render_mode, wants_padding=False, use_aspect=False, stretch=False):
    assert img.flags["C_CONTIGUOUS"] and img.dtype == np.uint8, "the descriptor takes the array's address: tightly packed RGB24"
    f = Frame()
    rc = lib().achip_frame_setup(C.byref(f), img.ctypes.data, img.shape[1], img.shape[0], width, height, render_mode,
                                 wants_padding, use_aspect, stretch)
    return f if rc == 0 else None


def frame_identity(img):
    assert img.flags["C_CONTIGUOUS"] and img.dtype == np.uint8
    f = Frame()
    assert lib().achip_frame_identity(C.byref(f), img.ctypes.data, img.shape[1], img.shape[0]) == 0
    return f


def crc32c_frames(frames, dims=None, stride=None, force=None, want_headers=True):
    """frames: list of bytes.  Returns (crc[n], headers[n] (24 B each) or None, packet_crc[n] or None) from the
    emulated wire-stage kernels.  force = (parts, rounds) overrides the launcher's span geometry."""
    n = len(frames)
    mx = max([len(f) for f in frames] + [1])
    if stride is None:
        stride = (mx + 15) & ~15
    slab = np.full(n * stride + 16, 0xEE, dtype=np.uint8)  # garbage behind every frame must not matter
    base = slab.ctypes.data + (-slab.ctypes.data % 16)
    view = np.ctypeslib.as_array((C.c_uint8 * (n * stride)).from_address(base))
    for i, f in enumerate(frames):
        view[i * stride:i * stride + len(f)] = np.frombuffer(f, dtype=np.uint8)
    ln = np.array([len(f) for f in frames], dtype=np.uint32)
    crc = np.zeros(n, dtype=np.uint32)
    hdr = np.zeros(n * 24, dtype=np.uint8)
    pkt = np.zeros(n, dtype=np.uint32)
    d = np.array(dims if dims is not None else [(0, 0)] * n, dtype=np.uint32).reshape(n, 2)
    fp, fr = force if force else (0, 0)
    lib().emu_crc32c(base, stride, ln.ctypes.data, 0, mx, n, fp, fr, d.ctypes.data if dims is not None else None,
                     crc.ctypes.data, hdr.ctypes.data if want_headers else None, pkt.ctypes.data if want_headers else None)
    if not want_headers:
        return crc, None, None
    return crc, [hdr[24 * i:24 * i + 24].tobytes() for i in range(n)], pkt
