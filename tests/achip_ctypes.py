"""ctypes mirrors of ascii-chat_amd/csrc/achip_types.h + helpers shared by the emulator and GPU tests."""
import ctypes as C

MODE_MONO, MODE_TRUE_FG, MODE_256_FG, MODE_16_FG, MODE_TRUE_BG = 0, 1, 2, 3, 4
MODE_HB_TRUE, MODE_HB_256, MODE_HB_16, MODE_HB_MONO = 5, 6, 7, 8
MODE_16_DITHER_BG = 9
ALL_MODES = list(range(10))
MODE_NAMES = ["mono", "true_fg", "256_fg", "16_fg", "true_bg", "hb_true", "hb_256", "hb_16", "hb_mono", "16_dither_bg"]

# (color_level, render_mode) that image_print_with_capabilities maps to each mode; TRUE_BG is only
# reachable through image_print_color_background() itself
MODE_CAPS = {MODE_MONO: (0, 0), MODE_TRUE_FG: (3, 0), MODE_256_FG: (2, 0), MODE_16_FG: (1, 0),
             MODE_HB_TRUE: (3, 2), MODE_HB_256: (2, 2), MODE_HB_16: (1, 2), MODE_HB_MONO: (0, 2),
             MODE_16_DITHER_BG: (3, 1)}

LEN_OVERFLOW = 0xFFFFFFFF
LEN_BADDESC = 0xFFFFFFFE


class CompSrc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("src_w", C.c_int32), ("src_h", C.c_int32), ("src_stride", C.c_int32), ("_pad0", C.c_int32), ("tile_w", C.c_int32),
                ("tile_h", C.c_int32), ("org_x", C.c_int32), ("org_y", C.c_int32), ("x_ratio", C.c_uint32),
                ("y_ratio", C.c_uint32)]


class Composite(C.Structure):
    _fields_ = [("canvas_w", C.c_int32), ("canvas_h", C.c_int32), ("cols", C.c_int32), ("rows", C.c_int32),
                ("cell_w", C.c_int32), ("cell_h", C.c_int32), ("n_src", C.c_int32), ("_pad", C.c_int32),
                ("s", CompSrc * 9)]


class Frame(C.Structure):
    _fields_ = [("src", C.c_void_p), ("comp", C.c_void_p), ("src_w", C.c_int32), ("src_h", C.c_int32),
                ("out_w", C.c_int32), ("out_h", C.c_int32), ("pad_left", C.c_int32), ("pad_top", C.c_int32),
                ("x_ratio", C.c_uint32), ("y_ratio", C.c_uint32), ("src_stride", C.c_int32), ("ops", C.c_uint32)]


class Uniform(C.Structure):
    _fields_ = [("f", Frame), ("src_pitch", C.c_int64), ("enabled", C.c_uint32), ("_pad", C.c_uint32)]


class Lut(C.Structure):
    _fields_ = [("glyph", C.c_uint32 * 256), ("glyph64", C.c_uint32 * 64), ("ramp", C.c_uint8 * 64), ("flags", C.c_uint32)]


def bind_host(L):
    """Declare the achip_host.h functions on a loaded library."""
    ss = C.c_ssize_t
    L.aspect_ratio.restype = None
    L.aspect_ratio.argtypes = [ss, ss, ss, ss, C.c_bool, C.POINTER(ss), C.POINTER(ss)]
    L.achip_lut_build.restype = C.c_int
    L.achip_lut_build.argtypes = [C.c_char_p, C.POINTER(Lut)]
    L.achip_mode_from_caps.restype = C.c_int
    L.achip_mode_from_caps.argtypes = [C.c_int, C.c_int]
    L.achip_frame_setup.restype = C.c_int
    L.achip_frame_setup.argtypes = [C.POINTER(Frame), C.c_void_p, C.c_int, C.c_int, ss, ss, C.c_int, C.c_bool,
                                    C.c_bool, C.c_bool]
    L.achip_frame_identity.restype = C.c_int
    L.achip_frame_identity.argtypes = [C.POINTER(Frame), C.c_void_p, C.c_int, C.c_int]
    L.achip_frame_set_display_ops.restype = C.c_int
    L.achip_frame_set_display_ops.argtypes = [C.POINTER(Frame), C.c_bool, C.c_bool, C.c_int]
    L.achip_rainbow_color.restype = None
    L.achip_rainbow_color.argtypes = [C.c_float, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.POINTER(C.c_uint8)]
    L.achip_frame_set_rainbow.restype = C.c_int
    L.achip_frame_set_rainbow.argtypes = [C.POINTER(Frame), C.c_float]
    L.achip_frames_uniform.restype = C.c_int
    L.achip_frames_uniform.argtypes = [C.POINTER(Frame), C.c_int, C.POINTER(Uniform)]
    L.achip_frame_set_dither_style.restype = C.c_int
    L.achip_frame_set_dither_style.argtypes = [C.POINTER(Frame), C.c_bool, C.c_bool]
    L.achip_nn_ratio.restype = C.c_uint32
    L.achip_nn_ratio.argtypes = [C.c_int, C.c_int]
    L.achip_out_bound.restype = C.c_size_t
    L.achip_out_bound.argtypes = [C.c_int, C.POINTER(Frame)]
    L.achip_choose_geometry.restype = C.c_int
    L.achip_choose_geometry.argtypes = [C.c_int, C.POINTER(Frame), C.c_int, C.c_bool, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int,
                                        C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.achip_palette_ascii_only.restype = C.c_bool
    L.achip_palette_ascii_only.argtypes = [C.c_char_p]
    L.achip_frame_blob_parse.restype = C.c_int
    L.achip_frame_blob_parse.argtypes = [C.c_char_p, C.c_size_t, C.c_bool, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                         C.POINTER(C.c_void_p)]
    L.achip_grid_layout.restype = None
    L.achip_grid_layout.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int,
                                    C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.achip_composite_setup.restype = None
    L.achip_composite_setup.argtypes = [C.POINTER(Composite), C.POINTER(C.c_void_p), C.POINTER(C.c_int),
                                        C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int]
    return L
