"""A THIRD, independent statement of the render path's byte grammar -- test infrastructure only.

Why it exists ("what's weak" 1): the product's host logic and the C oracle were written by the same hand,
often as the same lines, so "product == oracle" says less than it seems.  This module restates SURVEY.md section 8(a)
(rows R1, A1, L1-L3, PM, PT, P256, P16, PD, HT, H256, H16, HM, O1, W1, W2, E2, E3/D1) a third time, in another language
and with another structure: it builds Python lists of byte strings row by row with explicit sequential state -- the
shape of the reference's emitters -- instead of the oracle's sink / token machinery.  It was written from the survey's
text plus the reference files the survey cites for the two float formulas (lib/util/aspect_ratio.c:18-67) and the
16-colour table (lib/video/terminal/ansi.c:442-459), without looking at oracle/asciichat_oracle.c.
tests/test_third_restatement.py cross-checks it against the C oracle byte for byte.

Pure Python + numpy, small inputs only (seconds for a 200x60 grid).
"""
import numpy as np

ESC = b"\033"
RESET = b"\033[0m"
F32 = np.float32

ANSI16 = [(0, 0, 0), (128, 0, 0), (0, 128, 0), (128, 128, 0), (0, 0, 128), (128, 0, 128), (0, 128, 128), (192, 192, 192),
          (128, 128, 128), (255, 0, 0), (0, 255, 0), (255, 255, 0), (0, 0, 255), (255, 0, 255), (0, 255, 255),
          (255, 255, 255)]


# ---- R1 / A1 ----------------------------------------------------------------------------------------------
def resize_nearest(img, dw, dh):
    """image_resize: 16.16 ratios ((src << 16) / dst) + 1, index (i * ratio) >> 16 clamped to src - 1."""
    sh, sw = img.shape[:2]
    xr, yr = ((sw << 16) // dw) + 1, ((sh << 16) // dh) + 1
    xs = [min(((x * xr) & 0xFFFFFFFF) >> 16, sw - 1) for x in range(dw)]
    ys = [min(((y * yr) & 0xFFFFFFFF) >> 16, sh - 1) for y in range(dh)]
    return img[np.array(ys)][:, np.array(xs)]


def _round(x):  # ROUND(x) = (int)(0.5f + x), float32
    return int(F32(0.5) + F32(x))


def fit(img_w, img_h, width, height, stretch=False):
    """aspect_ratio(): terminal cells are twice as tall as wide (CHAR_ASPECT 2.0f); float32 throughout."""
    if img_w <= 0 or img_h <= 0:
        return 1, 1
    if stretch:
        return width, height
    w_from_h = _round(F32(height) * F32(img_w) / F32(img_h) * F32(2.0))
    w_from_h = w_from_h if w_from_h > 0 else 1
    h_from_w = _round((F32(width) / F32(2.0)) * F32(img_h) / F32(img_w))
    h_from_w = h_from_w if h_from_w > 0 else 1
    ow, oh = (w_from_h, height) if w_from_h <= width else (width, h_from_w)
    return max(ow, 1), max(oh, 1)


# ---- L1-L3, O1 ---------------------------------------------------------------------------------------------
def luma(p):
    return (77 * int(p[0]) + 150 * int(p[1]) + 29 * int(p[2]) + 128) >> 8


def split_palette(palette):
    """UTF-8 characters by the lead-byte rule (common.c:397-410), at most 255 of them."""
    b = palette.encode("utf-8") if isinstance(palette, str) else bytes(palette)
    out, i = [], 0
    while i < len(b) and len(out) < 255:
        c = b[i]
        n = 2 if c & 0xE0 == 0xC0 else 3 if c & 0xF0 == 0xE0 else 4 if c & 0xF8 == 0xF0 else 1
        out.append(b[i:i + n])
        i += n
    return out


def glyph_tables(palette):
    ch = split_palette(palette)
    n = len(ch)
    cache = [ch[(i * (n - 1) + 127) // 255] for i in range(256)]
    ramp = [(i * (n - 1) + 31) // 63 for i in range(64)]
    cache64 = [ch[ramp[i]] for i in range(64)]
    return cache, ramp, cache64


def ndigits(v):
    return len(str(v))


def rep_pays(run):
    return run > 2 and (run - 1) > ndigits(run - 1) + 3


def run_tail(glyph, run):
    """What follows the first glyph of a run of `run` equal cells."""
    return ESC + b"[%db" % (run - 1) if rep_pays(run) else glyph * (run - 1)


def to256(p):
    r, g, b = int(p[0]), int(p[1]), int(p[2])
    avg = (r + g + b) // 3
    if abs(r - avg) + abs(g - avg) + abs(b - avg) < 30:
        return 232 + avg * 23 // 255
    return 16 + 36 * (r * 5 // 255) + 6 * (g * 5 // 255) + (b * 5 // 255)


def to16(p):
    r, g, b = int(p[0]), int(p[1]), int(p[2])
    best, best_d = 0, None
    for i, (cr, cg, cb) in enumerate(ANSI16):
        d = (r - cr) ** 2 + (g - cg) ** 2 + (b - cb) ** 2
        if best_d is None or d < best_d:
            best, best_d = i, d
    return best


def sgr_true(bg, p):
    return ESC + b"[%d8;2;%d;%d;%dm" % (4 if bg else 3, int(p[0]), int(p[1]), int(p[2]))


def sgr_256(bg, idx):
    return ESC + b"[%d8;5;%dm" % (4 if bg else 3, idx)


def sgr_16(bg, idx):
    base = (40 if idx < 8 else 92) if bg else (30 if idx < 8 else 82)
    return ESC + b"[%dm" % (base + idx)


def rows_of_runs(keys):
    """[(start, length)] of maximal runs of equal consecutive keys in one row."""
    runs, x = [], 0
    while x < len(keys):
        e = x + 1
        while e < len(keys) and keys[e] == keys[x]:
            e += 1
        runs.append((x, e - x))
        x = e
    return runs


# ---- the ten renderers on an already-sized image ------------------------------------------------------------
def mono(img, palette):  # PM, with the double-mapped glyph index
    _, ramp, cache64 = glyph_tables(palette)
    lines = []
    for row in img:
        keys = [ramp[luma(p) >> 2] for p in row]
        out = b""
        for x, run in rows_of_runs(keys):
            g = cache64[min(keys[x], 63)]
            out += g + run_tail(g, run)
        lines.append(out)
    return b"\n".join(lines)


def truecolor_fg(img, palette):  # PT
    cache, _, _ = glyph_tables(palette)
    have, last = False, None
    lines = []
    for row in img:
        out = b""
        for p in row:
            g = cache[luma(p)]
            rgb = (int(p[0]), int(p[1]), int(p[2]))
            if len(g) == 1 and g[0] < 128:
                if not have or rgb != last:
                    out += sgr_true(False, p)
                have, last = True, rgb
            else:  # multi-byte glyph: always an SGR, the colour state is left alone
                out += sgr_true(False, p)
            out += g
        lines.append(out)
    return b"\n".join(lines) + RESET


def ansi256_fg(img, palette):  # P256
    cache, _, _ = glyph_tables(palette)
    return b"\n".join(b"".join(sgr_256(False, to256(p)) + cache[luma(p)] for p in row) + RESET for row in img)


def ansi16_fg(img, palette):  # P16, glyph through cache[ramp[Y >> 2]] (sic)
    cache, ramp, _ = glyph_tables(palette)
    return b"\n".join(b"".join(sgr_16(False, to16(p)) + cache[ramp[luma(p) >> 2]] for p in row) + RESET for row in img)


def truecolor_bg(img, palette):  # PB
    cache, _, _ = glyph_tables(palette)
    lines = []
    for row in img:
        out = b""
        for p in row:
            y = luma(p)
            out += sgr_true(True, p) + sgr_true(False, (255, 255, 255) if y < 128 else (0, 0, 0)) + cache[y]
        lines.append(out + RESET)
    return b"\n".join(lines)


def _c_div(a, b):  # C integer division truncates toward zero
    return -((-a) // b) if a < 0 else a // b


def dither16_bg(img, palette):  # PD, the form TRUECOLOR + BACKGROUND dispatches to
    cache, _, _ = glyph_tables(palette)
    h, w = img.shape[:2]
    err = [[[0, 0, 0] for _ in range(w)] for _ in range(h)]
    lines = []
    for y in range(h):
        out = b""
        for x in range(w):
            p = img[y, x]
            v = [int(p[k]) + err[y][x][k] for k in range(3)]
            err[y][x] = [0, 0, 0]
            idx = to16([min(255, max(0, c)) for c in v])
            e = [v[k] - ANSI16[idx][k] for k in range(3)]  # from the UNclamped value
            for (dx, dy, wgt) in ((1, 0, 7), (-1, 1, 3), (0, 1, 5), (1, 1, 1)):
                if 0 <= x + dx < w and y + dy < h:
                    for k in range(3):
                        err[y + dy][x + dx][k] += _c_div(e[k] * wgt, 16)
            pr, pg, pb = ANSI16[idx]
            bright = (77 * pr + 150 * pg + 29 * pb) // 256
            out += sgr_16(True, idx) + (ESC + b"[97m" if bright < 127 else ESC + b"[30m") + cache[luma(p)]
        lines.append(out + RESET)
    return b"\n".join(lines)


def _pairs(img):
    """Half-block rows: (top row, bottom row); an odd last row repeats its top."""
    h = img.shape[0]
    return [(img[y], img[y + 1] if y + 1 < h else img[y]) for y in range(0, h, 2)]


def _halfblock(img, key_of, sgr_of):
    """HT / H256 / H16: run membership on (key(top), key(bottom)); transparency on the run head's RAW rgb."""
    block = "▀".encode()
    lines = []
    for top, bot in _pairs(img):
        keys = [(key_of(t), key_of(b)) for t, b in zip(top, bot)]
        out, fg, bg = b"", None, None
        for x, run in rows_of_runs(keys):
            if not top[x].any() and not bot[x].any():
                if fg is not None or bg is not None:
                    out += RESET
                    fg = bg = None
                out += b" " * run
                continue
            kf, kb = keys[x]
            if fg != kf:
                out += sgr_of(False, top[x], kf)
                fg = kf
            if bg != kb:
                out += sgr_of(True, bot[x], kb)
                bg = kb
            out += block + run_tail(block, run)
        lines.append(out + RESET)
    return b"\n".join(lines)


def halfblock_true(img, palette=None):
    return _halfblock(img, lambda p: (int(p[0]), int(p[1]), int(p[2])), lambda bg, p, k: sgr_true(bg, p))


def halfblock_256(img, palette=None):
    return _halfblock(img, to256, lambda bg, p, k: sgr_256(bg, k))


def halfblock_16(img, palette=None):
    return _halfblock(img, to16, lambda bg, p, k: sgr_16(bg, k))


def halfblock_mono(img, palette=None):  # HM: 76/150/29 luminance without the rounding term
    shades = ["░".encode(), "▒".encode(), "▓".encode(), "█".encode()]
    lines = []
    for top, bot in _pairs(img):
        keys = [(tuple(int(c) for c in t), tuple(int(c) for c in b)) for t, b in zip(top, bot)]
        out = b""
        for x, run in rows_of_runs(keys):
            lt = (76 * int(top[x][0]) + 150 * int(top[x][1]) + 29 * int(top[x][2])) >> 8
            lb = (76 * int(bot[x][0]) + 150 * int(bot[x][1]) + 29 * int(bot[x][2])) >> 8
            if lt < 16 and lb < 16:
                out += b" " * run
            else:
                g = shades[lt >> 6]
                out += g + run_tail(g, run)
        lines.append(out)
    return b"\n".join(lines)


# ---- E3 / D1 dispatch, W1 / W2 padding, E2 sizing ------------------------------------------------------------
def print_with_caps(img, color_level, render_mode, palette):
    if render_mode == 2:
        return {3: halfblock_true, 2: halfblock_256, 1: halfblock_16}.get(color_level, halfblock_mono)(img, palette)
    if color_level == 3:
        return dither16_bg(img, palette) if render_mode == 1 else truecolor_fg(img, palette)
    if color_level == 2:
        return ansi256_fg(img, palette)
    if color_level == 1:
        return ansi16_fg(img, palette)
    return mono(img, palette)


def pad_width(frame, pad_left):
    return frame if pad_left == 0 else b"\n".join(b" " * pad_left + line for line in frame.split(b"\n"))


def pad_height(frame, pad_top):
    return b"\n" * pad_top + frame


def convert_with_caps(img, width, height, color_level, render_mode, wants_padding=False, use_aspect=False,
                      stretch=False, palette="   ...',;:clodxkO0KXNWM"):
    """ascii_convert_with_capabilities: aspect fit BEFORE half-block doubling; padding only when both flags are set."""
    sh, sw = img.shape[:2]
    rw, rh = (fit(sw, sh, width, height, stretch) if use_aspect else (width, height))
    out_w, out_h = rw, rh
    if render_mode == 2:
        rh *= 2
    pad_l = (width - out_w) // 2 if (use_aspect and wants_padding and width > out_w) else 0
    pad_t = (height - out_h) // 2 if (use_aspect and wants_padding and height > out_h) else 0
    frame = print_with_caps(resize_nearest(img, rw, rh), color_level, render_mode, palette)
    return pad_height(pad_width(frame, pad_l), pad_t)
