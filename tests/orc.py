"""ctypes binding to the CPU oracle (oracle/asciichat_oracle.c) + synthetic frame generators.

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "_build", "libasciichat_oracle.so")

COLOR_AUTO, COLOR_NONE, COLOR_16, COLOR_256, COLOR_TRUECOLOR = -1, 0, 1, 2, 3
RENDER_FG, RENDER_BG, RENDER_HALF_BLOCK = 0, 1, 2

PALETTE_STANDARD = "   ...',;:clodxkO0KXNWM"
PALETTE_BLOCKS = "   ░░▒▒▓▓██"
PALETTE_DIGITAL = "   -=≡≣▰▱◼"
PALETTE_MINIMAL = "   .-+*#"
PALETTE_COOL = "   ▁▂▃▄▅▆▇█"


def build_oracle(force=False):
    src = [os.path.join(ORACLE_DIR, f) for f in ("asciichat_oracle.c", "asciichat_oracle.h", "oracle_bench.c")]
    if force or not os.path.exists(ORACLE_SO) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(ORACLE_SO) for s in src
    ):
        if all(os.path.exists(s) for s in src):
            subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)
    return ORACLE_SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build_oracle())
        L = _lib
        vp, ci, cl, sz = C.c_void_p, C.c_int, C.c_long, C.c_size_t
        L.orc_fnv1a32.restype = C.c_uint32
        L.orc_fnv1a32.argtypes = [vp, sz]
        for f in (L.orc_expand_rle, L.orc_compress_rle):
            f.restype = vp
            f.argtypes = [C.c_char_p, sz, C.POINTER(sz)]
        L.orc_frame_validate_integrity.restype = C.c_int
        L.orc_frame_validate_integrity.argtypes = [C.c_char_p, sz]
        L.orc_frame_get_valid_end.restype = sz
        L.orc_frame_get_valid_end.argtypes = [C.c_char_p, sz]
        L.orc_frame_blob_accept.restype = C.c_int
        L.orc_frame_blob_accept.argtypes = [C.c_char_p, sz, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.orc_crc32c.restype = C.c_uint32
        L.orc_crc32c.argtypes = [C.c_char_p, sz]
        L.orc_ascii_frame_packet.restype = C.c_uint32
        L.orc_ascii_frame_packet.argtypes = [C.c_char_p, sz, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint8 * 24)]
        for name in ("orc_print_mono", "orc_print_truecolor_fg", "orc_print_256_fg", "orc_print_16_fg",
                     "orc_print_truecolor_bg"):
            f = getattr(L, name)
            f.restype = vp
            f.argtypes = [vp, ci, ci, C.c_char_p, C.POINTER(sz)]
        L.orc_print_16_dithered.restype = vp
        L.orc_print_16_dithered.argtypes = [vp, ci, ci, C.c_bool, C.c_char_p, C.POINTER(sz)]
        L.orc_rainbow_color.restype = None
        L.orc_rainbow_color.argtypes = [C.c_float, C.POINTER(C.c_uint8 * 3)]
        L.orc_rainbow_replace.restype = vp
        L.orc_rainbow_replace.argtypes = [C.c_char_p, C.c_float, C.POINTER(sz)]
        L.orc_print_16_dithered_fg.restype = vp
        L.orc_print_16_dithered_fg.argtypes = [vp, ci, ci, C.c_char_p, C.POINTER(sz)]
        for name in ("orc_halfblock_truecolor", "orc_halfblock_256", "orc_halfblock_16", "orc_halfblock_mono"):
            f = getattr(L, name)
            f.restype = vp
            f.argtypes = [vp, ci, ci, C.POINTER(sz)]
        L.orc_print_with_caps.restype = vp
        L.orc_print_with_caps.argtypes = [vp, ci, ci, ci, ci, C.c_char_p, C.POINTER(sz)]
        L.orc_convert_with_caps.restype = vp
        L.orc_convert_with_caps.argtypes = [vp, ci, ci, cl, cl, ci, ci, C.c_bool, C.c_bool, C.c_bool, C.c_char_p,
                                            C.POINTER(sz)]
        L.orc_convert.restype = vp
        L.orc_convert.argtypes = [vp, ci, ci, cl, cl, C.c_bool, C.c_bool, C.c_bool, C.c_char_p, ci, C.POINTER(sz)]
        L.orc_aspect_ratio.restype = None
        L.orc_aspect_ratio.argtypes = [cl, cl, cl, cl, C.c_bool, C.POINTER(cl), C.POINTER(cl)]
        L.orc_resize_nn.restype = None
        L.orc_resize_nn.argtypes = [vp, ci, ci, vp, ci, ci]
        L.orc_rgb_to_256.restype = C.c_uint8
        L.orc_rgb_to_256.argtypes = [C.c_uint8] * 3
        L.orc_rgb_to_16.restype = C.c_uint8
        L.orc_rgb_to_16.argtypes = [C.c_uint8] * 3
        L.orc_rep_is_profitable.restype = C.c_bool
        L.orc_rep_is_profitable.argtypes = [C.c_uint32]
        L.orc_digits_u32.restype = ci
        L.orc_digits_u32.argtypes = [C.c_uint32]
        for name in ("orc_sgr_truecolor",):
            f = getattr(L, name)
            f.restype = ci
            f.argtypes = [C.c_char_p, ci, C.c_uint8, C.c_uint8, C.c_uint8]
        for name in ("orc_sgr_256", "orc_sgr_16"):
            f = getattr(L, name)
            f.restype = ci
            f.argtypes = [C.c_char_p, ci, C.c_uint8]
        L.orc_pad_width.restype = vp
        L.orc_pad_width.argtypes = [C.c_char_p, sz]
        L.orc_pad_height.restype = vp
        L.orc_pad_height.argtypes = [C.c_char_p, sz]
        L.orc_create_grid.restype = vp
        L.orc_create_grid.argtypes = [vp, ci, ci, ci, C.POINTER(sz)]
        L.orc_grid_layout.restype = None
        L.orc_grid_layout.argtypes = [vp, vp, ci, ci, ci, C.POINTER(ci), C.POINTER(ci)]
        L.orc_composite.restype = vp
        L.orc_composite.argtypes = [vp, vp, vp, ci, ci, ci, C.POINTER(ci), C.POINTER(ci)]
        L.orc_flip.restype = None
        L.orc_flip.argtypes = [vp, ci, ci, C.c_bool, C.c_bool]
        L.orc_color_filter.restype = ci
        L.orc_color_filter.argtypes = [vp, ci, ci, ci, ci]
        L.orc_display_convert.restype = vp
        L.orc_display_convert.argtypes = [vp, ci, ci, cl, cl, ci, ci, C.c_bool, C.c_bool, C.c_bool, C.c_char_p,
                                          C.c_bool, C.c_bool, ci, C.POINTER(sz)]
        L.orc_bench_convert.restype = C.c_double
        L.orc_bench_convert.argtypes = [vp, ci, ci, ci, ci, ci, ci, C.c_char_p, ci, ci, C.POINTER(C.c_uint64)]
        L.free.argtypes = [vp]
    return _lib


def _take(ptr, n=None):
    """Copy a malloc'd C string result to bytes and free it. None for NULL."""
    if not ptr:
        return None
    out = C.string_at(ptr) if n is None else C.string_at(ptr, n)
    lib().free(ptr)
    return out


def _img(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    assert a.ndim == 3 and a.shape[2] == 3
    return a


def _pal(p):
    return p.encode("utf-8") if isinstance(p, str) else p


def expand_rle(b):
    n = C.c_size_t()
    p = lib().orc_expand_rle(bytes(b), len(b), C.byref(n))
    return _take(p, n.value) if p else None


def compress_rle(b):
    n = C.c_size_t()
    p = lib().orc_compress_rle(bytes(b), len(b), C.byref(n))
    return _take(p, n.value) if p else None


def frame_blob_accept(blob, exact=False):
    """-> (w, h) when the reference accepts the camera frame blob, else None."""
    w, h = C.c_uint32(), C.c_uint32()
    ok = lib().orc_frame_blob_accept(bytes(blob), len(blob), int(exact), C.byref(w), C.byref(h))
    return (w.value, h.value) if ok else None


def crc32c(b):
    return lib().orc_crc32c(bytes(b), len(b))


def ascii_frame_packet(frame, width, height):
    """-> (24-byte header, CRC of header+frame) as acip_send_ascii_frame / packet_send_via_transport produce them."""
    hdr = (C.c_uint8 * 24)()
    crc = lib().orc_ascii_frame_packet(bytes(frame), len(frame), width, height, hdr)
    return bytes(hdr), crc


def fnv1a32(b):
    return int(lib().orc_fnv1a32(b, len(b)))


def print_with_caps(img, color_level, render_mode, palette=PALETTE_STANDARD):
    img = _img(img)
    n = C.c_size_t()
    p = lib().orc_print_with_caps(img.ctypes.data, img.shape[1], img.shape[0], color_level, render_mode,
                                  _pal(palette), C.byref(n))
    return _take(p, n.value)


def print_16_dithered(img, use_background, palette=PALETTE_STANDARD, ramp_glyph=False):
    """image_print_16color_dithered_with_background(img, use_background, pal), or with ramp_glyph the
    foreground-only image_print_16color_dithered(img, pal)"""
    img = _img(img)
    n = C.c_size_t()
    if ramp_glyph:
        assert not use_background
        p = lib().orc_print_16_dithered_fg(img.ctypes.data, img.shape[1], img.shape[0], _pal(palette), C.byref(n))
    else:
        p = lib().orc_print_16_dithered(img.ctypes.data, img.shape[1], img.shape[0], use_background, _pal(palette),
                                        C.byref(n))
    return _take(p, n.value)


def rainbow_color(t):
    c = (C.c_uint8 * 3)()
    lib().orc_rainbow_color(t, C.byref(c))
    return tuple(c)


def rainbow_replace(frame, t):
    """rainbow_replace_ansi_colors as its callers use it: the frame itself when there is nothing to recolour"""
    assert b"\0" not in frame
    n = C.c_size_t()
    p = lib().orc_rainbow_replace(bytes(frame), t, C.byref(n))
    return _take(p, n.value) if p else bytes(frame)


def print_truecolor_bg(img, palette=PALETTE_STANDARD):
    img = _img(img)
    n = C.c_size_t()
    p = lib().orc_print_truecolor_bg(img.ctypes.data, img.shape[1], img.shape[0], _pal(palette), C.byref(n))
    return _take(p, n.value)


def convert_with_caps(img, width, height, color_level, render_mode, wants_padding=False, use_aspect=False,
                      stretch=False, palette=PALETTE_STANDARD):
    img = _img(img)
    n = C.c_size_t()
    p = lib().orc_convert_with_caps(img.ctypes.data, img.shape[1], img.shape[0], width, height, color_level,
                                    render_mode, wants_padding, use_aspect, stretch, _pal(palette), C.byref(n))
    return _take(p, n.value)


def convert(img, width, height, color, use_aspect, stretch, palette=PALETTE_STANDARD, option_render_mode=RENDER_FG):
    img = _img(img)
    n = C.c_size_t()
    p = lib().orc_convert(img.ctypes.data, img.shape[1], img.shape[0], width, height, color, use_aspect, stretch,
                          _pal(palette), option_render_mode, C.byref(n))
    return _take(p, n.value)


def display_convert(img, width, height, color_level, render_mode, wants_padding=False, use_aspect=False,
                    flip_x=False, flip_y=False, color_filter=0, palette=PALETTE_STANDARD):
    img = _img(img)
    n = C.c_size_t()
    p = lib().orc_display_convert(img.ctypes.data, img.shape[1], img.shape[0], width, height, color_level, render_mode,
                                  wants_padding, use_aspect, False, _pal(palette), flip_x, flip_y, color_filter,
                                  C.byref(n))
    return _take(p, n.value)


def color_filter(img, flt):
    out = _img(img).copy()
    assert lib().orc_color_filter(out.ctypes.data, out.shape[1], out.shape[0], out.shape[1] * 3, flt) == 0
    return out


def flip(img, flip_x, flip_y):
    out = _img(img).copy()
    lib().orc_flip(out.ctypes.data, out.shape[1], out.shape[0], flip_x, flip_y)
    return out


def aspect_ratio(img_w, img_h, width, height, stretch=False):
    ow, oh = C.c_long(), C.c_long()
    lib().orc_aspect_ratio(img_w, img_h, width, height, stretch, C.byref(ow), C.byref(oh))
    return ow.value, oh.value


def resize_nn(img, dw, dh):
    img = _img(img)
    out = np.zeros((dh, dw, 3), dtype=np.uint8)
    lib().orc_resize_nn(img.ctypes.data, img.shape[1], img.shape[0], out.ctypes.data, dw, dh)
    return out


class _FrameSource(C.Structure):
    _fields_ = [("frame_data", C.c_char_p), ("frame_size", C.c_size_t)]


def create_grid(frames, width, height):
    arr = (_FrameSource * max(1, len(frames)))()
    for i, f in enumerate(frames):
        arr[i].frame_data = f
        arr[i].frame_size = len(f) if f is not None else 0
    n = C.c_size_t()
    p = lib().orc_create_grid(arr, len(frames), width, height, C.byref(n))
    return _take(p, n.value)


def grid_layout(dims, term_w, term_h):
    ws = (C.c_int * len(dims))(*[d[0] for d in dims])
    hs = (C.c_int * len(dims))(*[d[1] for d in dims])
    c, r = C.c_int(), C.c_int()
    lib().orc_grid_layout(ws, hs, len(dims), term_w, term_h, C.byref(c), C.byref(r))
    return c.value, r.value


def composite(imgs, term_w, term_h):
    """imgs: list of HxWx3 arrays; None = a client without video (takes no cell, stream.c:690-692)"""
    imgs = [None if i is None else _img(i) for i in imgs]
    ptrs = (C.c_void_p * len(imgs))(*[None if i is None else i.ctypes.data for i in imgs])
    ws = (C.c_int * len(imgs))(*[0 if i is None else i.shape[1] for i in imgs])
    hs = (C.c_int * len(imgs))(*[0 if i is None else i.shape[0] for i in imgs])
    ow, oh = C.c_int(), C.c_int()
    p = lib().orc_composite(ptrs, ws, hs, len(imgs), term_w, term_h, C.byref(ow), C.byref(oh))
    out = np.frombuffer(C.string_at(p, ow.value * oh.value * 3), dtype=np.uint8).reshape(oh.value, ow.value, 3).copy()
    lib().free(p)
    return out


# --------------------------------------------------------------------------- #
# synthetic inputs; all uint8 HxWx3, tightly packed   #
# --------------------------------------------------------------------------- #
def xorshift32_stream(seed, n):
    out = np.empty(n, dtype=np.uint32)
    x = np.uint32(seed)
    v = int(x)
    for i in range(n):
        v ^= (v << 13) & 0xFFFFFFFF
        v ^= v >> 17
        v ^= (v << 5) & 0xFFFFFFFF
        out[i] = v
    return out


def _xorshift32_stream_fast(seed, n):
    # vector-free but chunked python is too slow for 4K; use a small C-like numpy loop over bit ops
    return xorshift32_stream(seed, n)


def frame_noise(w, h, seed=12345):
    """S-noise: one xorshift32 draw per pixel, R=v, G=v>>8, B=v>>16."""
    v = xorshift32_stream(seed, w * h)
    img = np.empty((h * w, 3), dtype=np.uint8)
    img[:, 0] = v & 0xFF
    img[:, 1] = (v >> 8) & 0xFF
    img[:, 2] = (v >> 16) & 0xFF
    return img.reshape(h, w, 3)


def frame_hash_noise(w, h, seed=1):
    """Fast counter-hash noise for large frames (not one of the survey's inputs; used where xorshift's
    sequential generator would dominate test time). Also implemented on device in bench.py."""
    i = np.arange(w * h, dtype=np.uint64) + np.uint64((seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
    i ^= i >> np.uint64(33)
    i *= np.uint64(0xFF51AFD7ED558CCD)
    i ^= i >> np.uint64(33)
    i *= np.uint64(0xC4CEB9FE1A85EC53)
    i ^= i >> np.uint64(33)
    img = np.empty((h * w, 3), dtype=np.uint8)
    img[:, 0] = (i & np.uint64(0xFF)).astype(np.uint8)
    img[:, 1] = ((i >> np.uint64(8)) & np.uint64(0xFF)).astype(np.uint8)
    img[:, 2] = ((i >> np.uint64(16)) & np.uint64(0xFF)).astype(np.uint8)
    return img.reshape(h, w, 3)


def frame_smooth(w, h):
    """S-smooth: R=x*255/(w-1), G=y*255/(h-1), B=((x/64)+(y/64))*32."""
    x = np.arange(w, dtype=np.int64)[None, :]
    y = np.arange(h, dtype=np.int64)[:, None]
    img = np.empty((h, w, 3), dtype=np.uint8)
    img[..., 0] = np.broadcast_to(x * 255 // max(1, w - 1), (h, w))
    img[..., 1] = np.broadcast_to(y * 255 // max(1, h - 1), (h, w))
    img[..., 2] = (((x // 64) + (y // 64)) * 32) & 0xFF
    return img


def frame_bars(w, h, frame_index=0):
    """S-bars: the reference's webcam test pattern (lib/video/webcam/webcam.c:78-117) scaled to the
    frame: 3-colour bars of width w/8 with black grid lines, phase = frame_index/2."""
    bar = max(1, w // 8)
    rowgap = max(1, h // 8)
    phase = frame_index // 2
    x = np.arange(w, dtype=np.int64)[None, :]
    y = np.arange(h, dtype=np.int64)[:, None]
    ax = (x + phase) % w
    sel = (ax // bar) % 3
    img = np.zeros((h, w, 3), dtype=np.uint8)
    for c in range(3):
        img[..., c] = np.broadcast_to(np.where(sel == c, 255, 0), (h, w))
    grid = np.broadcast_to((ax % bar == 0), (h, w)) | np.broadcast_to((y % rowgap == 0), (h, w))
    img[grid] = 0
    return img


def frame_gray(w, h):
    """S-gray: R=G=B=(i*255)/(w*h) (tests/unit/video/ascii_test.c:49-52 in the reference)."""
    i = np.arange(w * h, dtype=np.int64)
    g = (i * 255 // (w * h)).astype(np.uint8)
    return np.repeat(g[:, None], 3, axis=1).reshape(h, w, 3)


def frame_anchor_gradient(w=640, h=480):
    """SURVEY 8(c) sanity-anchor input: px(x,y) = (x*255/639, y*255/479, (x+y)&255)."""
    x = np.arange(w, dtype=np.int64)[None, :]
    y = np.arange(h, dtype=np.int64)[:, None]
    img = np.empty((h, w, 3), dtype=np.uint8)
    img[..., 0] = np.broadcast_to(x * 255 // (w - 1), (h, w))
    img[..., 1] = np.broadcast_to(y * 255 // (h - 1), (h, w))
    img[..., 2] = (x + y) & 255
    return img


def frame_torture(w=333, h=201):
    """SURVEY Appendix B 'torture' image: transparent runs, REP runs, gradient, noise."""
    img = np.zeros((h, w, 3), dtype=np.uint8)
    x = np.arange(w, dtype=np.int64)
    img[0:40, x > 200, :] = 1
    v = (x // 37) * 40
    img[40:90, :, 0] = (v & 0xFF)[None, :]
    img[40:90, :, 1] = ((255 - v) & 0xFF)[None, :]
    img[40:90, :, 2] = ((v // 2) & 0xFF)[None, :]
    yy = np.arange(90, 150, dtype=np.int64)[:, None]
    img[90:150, :, 0] = np.broadcast_to((x * 255 // (w - 1))[None, :], (60, w))
    img[90:150, :, 1] = np.broadcast_to(yy & 0xFF, (60, w))
    img[90:150, :, 2] = (x[None, :] + yy) & 255
    nrows = h - 150
    vv = xorshift32_stream(777, nrows * w)
    img[150:, :, 0] = (vv & 0xFF).reshape(nrows, w)
    img[150:, :, 1] = ((vv >> 8) & 0xFF).reshape(nrows, w)
    img[150:, :, 2] = ((vv >> 16) & 0xFF).reshape(nrows, w)
    return img
