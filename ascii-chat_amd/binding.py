"""ctypes binding of libasciichat_hip.so (C-ABI: include/asciichat_hip.h, include/asciichat_render.h)."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ASCIICHAT_HIP_LIB") or os.path.join(HERE, "libasciichat_hip.so")

MODE_MONO, MODE_TRUE_FG, MODE_256_FG, MODE_16_FG, MODE_TRUE_BG = 0, 1, 2, 3, 4
MODE_HB_TRUE, MODE_HB_256, MODE_HB_16, MODE_HB_MONO = 5, 6, 7, 8
MODE_16_DITHER_BG = 9
MODE_NAMES = ["mono", "true_fg", "256_fg", "16_fg", "true_bg", "hb_true", "hb_256", "hb_16", "hb_mono", "16_dither_bg"]
LEN_OVERFLOW, LEN_BADDESC = 0xFFFFFFFF, 0xFFFFFFFE
ERR_NO_DEVICE = 200


class CompSrc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("src_w", C.c_int32), ("src_h", C.c_int32), ("src_stride", C.c_int32),
                ("_pad0", C.c_int32), ("tile_w", C.c_int32), ("tile_h", C.c_int32), ("org_x", C.c_int32),
                ("org_y", C.c_int32), ("x_ratio", C.c_uint32), ("y_ratio", C.c_uint32)]


class Composite(C.Structure):
    _fields_ = [("canvas_w", C.c_int32), ("canvas_h", C.c_int32), ("cols", C.c_int32), ("rows", C.c_int32),
                ("cell_w", C.c_int32), ("cell_h", C.c_int32), ("n_src", C.c_int32), ("_pad", C.c_int32),
                ("s", CompSrc * 9)]


class Frame(C.Structure):
    _fields_ = [("src", C.c_void_p), ("comp", C.c_void_p), ("src_w", C.c_int32), ("src_h", C.c_int32),
                ("out_w", C.c_int32), ("out_h", C.c_int32), ("pad_left", C.c_int32), ("pad_top", C.c_int32),
                ("x_ratio", C.c_uint32), ("y_ratio", C.c_uint32), ("src_stride", C.c_int32), ("ops", C.c_uint32)]


class Image(C.Structure):  # image_t
    _fields_ = [("w", C.c_int), ("h", C.c_int), ("pixels", C.c_void_p), ("alloc_method", C.c_uint8)]


class TermCaps(C.Structure):  # terminal_capabilities_t
    _fields_ = [("color_level", C.c_int), ("capabilities", C.c_uint32), ("color_count", C.c_uint32),
                ("utf8_support", C.c_bool), ("detection_reliable", C.c_bool), ("render_mode", C.c_int),
                ("term_type", C.c_char * 64), ("colorterm", C.c_char * 64), ("wants_background", C.c_bool),
                ("palette_type", C.c_int), ("palette_custom", C.c_char * 64), ("desired_fps", C.c_uint8),
                ("color_filter", C.c_int), ("wants_padding", C.c_bool), ("pad_height", C.c_size_t)]


class FrameSource(C.Structure):  # ascii_frame_source_t
    _fields_ = [("frame_data", C.c_char_p), ("frame_size", C.c_size_t)]


def build(force=False):
    """Compile the library in-tree (hipcc --offload-arch=gfx950 via ascii-chat_amd/Makefile)."""
    if force:
        subprocess.check_call(["make", "-C", HERE, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", HERE], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    """Load libasciichat_hip.so (never a substitute: raises if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # torch bundles its own libamdhip64.so (soname libamdhip64.so.7): load it FIRST so that our library's
        # DT_NEEDED libamdhip64.so.7 binds to the same runtime instead of a second copy from /opt/rocm
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc, gfx950) first")
    L = C.CDLL(LIB_PATH)
    # ASCIICHAT_HIP_LIB names another build of THIS library (A/B runs against an older commit's .so): entry points that
    # build does not have yet are skipped while binding and raise when called
    _lib = _bind(_OlderBuild(L) if os.environ.get("ASCIICHAT_HIP_LIB") else L)
    return _lib


class _OlderBuild:
    def __init__(self, L):
        object.__setattr__(self, "_L", L)

    def __getattr__(self, name):
        try:
            return getattr(self._L, name)
        except AttributeError:
            def missing(*a, **k):
                raise RuntimeError(f"{LIB_PATH} has no {name}")
            object.__setattr__(self, name, missing)
            return missing

    def __setattr__(self, name, value):
        setattr(self._L, name, value)


def _bind(L):
    """argument and result types of every entry point the classes below call"""
    vp, ci, ss, sz = C.c_void_p, C.c_int, C.c_ssize_t, C.c_size_t
    L.asciichat_hip_device_count.restype = ci
    L.asciichat_hip_last_error.restype = C.c_char_p
    L.asciichat_hip_plan_create.restype = ci
    L.asciichat_hip_plan_create.argtypes = [C.POINTER(vp), ci, C.c_char_p, C.POINTER(Frame), ci]
    L.asciichat_hip_plan_update.restype = ci
    L.asciichat_hip_plan_update.argtypes = [vp, C.POINTER(Frame), vp]
    L.asciichat_hip_plan_out_stride.restype = sz
    L.asciichat_hip_plan_out_stride.argtypes = [vp]
    L.asciichat_hip_plan_get_exact_length.restype = ci
    L.asciichat_hip_plan_get_exact_length.argtypes = [vp]
    L.asciichat_hip_plan_set_exact_length.restype = ci
    L.asciichat_hip_plan_set_exact_length.argtypes = [vp, ci]
    L.asciichat_hip_plan_set_variant.restype = ci
    L.asciichat_hip_plan_set_variant.argtypes = [vp, ci]
    L.asciichat_hip_plan_get_variant.restype = ci
    L.asciichat_hip_plan_get_variant.argtypes = [vp]
    L.asciichat_hip_plan_set_split.restype = ci
    L.asciichat_hip_plan_set_split.argtypes = [vp, ci]
    L.asciichat_hip_plan_set_concurrency.restype = ci
    L.asciichat_hip_plan_set_concurrency.argtypes = [vp, ci]
    L.asciichat_hip_plan_set_uniform.restype = ci
    L.asciichat_hip_plan_set_uniform.argtypes = [vp, ci]
    L.asciichat_hip_plan_get_uniform.restype = ci
    L.asciichat_hip_plan_get_uniform.argtypes = [vp]
    L.achip_frames_uniform.restype = ci
    L.achip_frames_uniform.argtypes = [C.POINTER(Frame), ci, C.c_void_p]
    L.asciichat_hip_plan_get_parts.restype = ci
    L.asciichat_hip_plan_get_parts.argtypes = [vp]
    L.asciichat_hip_frame_table_create.restype = ci
    L.asciichat_hip_frame_table_create.argtypes = [C.POINTER(vp), ci]
    L.asciichat_hip_frame_table_destroy.restype = None
    L.asciichat_hip_frame_table_destroy.argtypes = [vp]
    L.asciichat_hip_frame_table_publish.restype = ci
    L.asciichat_hip_frame_table_publish.argtypes = [vp, ci, C.c_char_p, C.c_size_t, vp]
    L.asciichat_hip_frame_table_publish_rows.restype = ci
    L.asciichat_hip_frame_table_publish_rows.argtypes = [vp, ci, vp, C.c_size_t, C.POINTER(Frame), ci, vp]
    L.asciichat_hip_frame_table_publish_rows_batch.restype = ci
    L.asciichat_hip_frame_table_publish_rows_batch.argtypes = [vp, C.POINTER(ci), C.POINTER(vp), C.POINTER(C.c_size_t), ci,
                                                               C.POINTER(Frame), ci, vp]
    L.asciichat_hip_frame_table_latest.restype = ci
    L.asciichat_hip_frame_table_latest.argtypes = [vp, ci, vp, C.POINTER(vp), C.POINTER(ci), C.POINTER(ci),
                                                   C.POINTER(C.c_uint64)]
    L.asciichat_hip_frame_table_forget_stream.restype = None
    L.asciichat_hip_frame_table_stage.restype = ci
    L.asciichat_hip_frame_table_stage.argtypes = [vp, ci, vp, sz, C.POINTER(Frame)]
    L.asciichat_hip_frame_table_commit.restype = ci
    L.asciichat_hip_frame_table_commit.argtypes = [vp, vp]
    L.asciichat_hip_frame_table_publish_sampled_batch.restype = ci
    L.asciichat_hip_frame_table_publish_sampled_batch.argtypes = [vp, C.POINTER(ci), C.POINTER(vp), C.POINTER(C.c_size_t), ci,
                                                                  C.POINTER(Frame), ci, vp]
    L.asciichat_hip_ingest_threads.restype = ci
    L.asciichat_hip_frame_table_latest_frames.restype = ci
    L.asciichat_hip_frame_table_latest_frames.argtypes = [vp, C.POINTER(ci), ci, vp, C.POINTER(Frame)]
    L.asciichat_hip_frame_table_forget_stream.argtypes = [vp, vp]
    L.asciichat_hip_crc32c.restype = ci
    L.asciichat_hip_crc32c.argtypes = [vp, C.c_size_t, vp, C.c_uint32, C.c_uint32, ci, vp, vp]
    L.asciichat_hip_frame_packets.restype = ci
    L.asciichat_hip_frame_packets.argtypes = [vp, C.c_size_t, vp, C.c_uint32, ci, vp, vp, vp, vp, vp]
    L.asciichat_hip_frame_packets_packed.restype = ci
    L.asciichat_hip_frame_packets_packed.argtypes = [vp, C.c_size_t, vp, C.c_uint32, ci, vp, vp, vp, vp, vp, C.c_size_t, vp, vp, vp]
    L.asciichat_hip_plan_render_packets_packed.restype = ci
    L.asciichat_hip_plan_render_packets_packed.argtypes = [vp, vp, C.c_size_t, vp, vp, vp, vp, vp, vp, C.c_size_t, vp, vp, vp]
    L.asciichat_hip_plan_render.restype = ci
    L.asciichat_hip_plan_render.argtypes = [vp, vp, sz, vp, vp]
    L.asciichat_hip_plan_render_range.restype = ci
    L.asciichat_hip_plan_render_range.argtypes = [vp, ci, ci, vp, sz, vp, vp]
    L.asciichat_hip_plan_render_profiled.restype = ci
    L.asciichat_hip_plan_render_profiled.argtypes = [vp, vp, sz, vp, vp, vp]
    L.asciichat_hip_plan_destroy.restype = None
    L.asciichat_hip_plan_destroy.argtypes = [vp]
    L.asciichat_hip_render_many_profiled.restype = ci
    L.asciichat_hip_render_many_profiled.argtypes = [C.POINTER(vp), ci, C.POINTER(vp), C.POINTER(vp), sz, C.POINTER(vp), ci,
                                                     ci, ci, vp, sz]
    L.asciichat_hip_render_many.restype = ci
    L.asciichat_hip_render_many.argtypes = [C.POINTER(vp), ci, C.POINTER(vp), C.POINTER(vp), sz, C.POINTER(vp), ci, ci, ci]
    L.asciichat_hip_streams_wait.restype = ci
    L.asciichat_hip_streams_wait.argtypes = [C.POINTER(vp), ci]
    L.asciichat_hip_comm_unique_id.restype = ci
    L.asciichat_hip_comm_unique_id.argtypes = [vp, sz]
    L.asciichat_hip_comm_init.restype = ci
    L.asciichat_hip_comm_init.argtypes = [C.POINTER(vp), ci, ci, vp, sz]
    L.asciichat_hip_comm_world.restype = ci
    L.asciichat_hip_comm_world.argtypes = [vp]
    L.asciichat_hip_comm_rank.restype = ci
    L.asciichat_hip_comm_rank.argtypes = [vp]
    L.asciichat_hip_comm_count.restype = ci
    L.asciichat_hip_comm_count.argtypes = [vp]
    L.asciichat_hip_comm_all_gather_packed.restype = ci
    L.asciichat_hip_comm_all_gather_packed.argtypes = [vp, vp, sz, vp, ci, vp, sz, C.POINTER(C.c_uint64),
                                                       C.POINTER(C.c_uint32), C.POINTER(sz), vp]
    L.asciichat_hip_pack_frames.restype = ci
    L.asciichat_hip_pack_frames.argtypes = [vp, sz, vp, ci, vp, sz, vp, vp, vp]
    L.asciichat_hip_plan_render_packed.restype = ci
    L.asciichat_hip_plan_render_packed.argtypes = [vp, vp, sz, vp, vp, sz, vp, vp, vp]
    L.asciichat_hip_host_alloc.restype = ci
    L.asciichat_hip_host_alloc.argtypes = [sz, C.POINTER(vp), C.POINTER(vp)]
    L.asciichat_hip_host_free.restype = None
    L.asciichat_hip_host_free.argtypes = [vp]
    L.asciichat_hip_comm_destroy.restype = None
    L.asciichat_hip_comm_destroy.argtypes = [vp]
    L.asciichat_hip_comm_all_gather.restype = ci
    L.asciichat_hip_comm_all_gather.argtypes = [vp, vp, vp, sz, vp]
    L.asciichat_hip_comm_all_gather_slab.restype = ci
    L.asciichat_hip_comm_all_gather_slab.argtypes = [vp, vp, sz, vp, ci, vp]
    L.achip_shard_bounds.restype = None
    L.achip_shard_bounds.argtypes = [ci, ci, ci, C.POINTER(ci), C.POINTER(ci)]
    L.achip_shard_owner.restype = ci
    L.achip_shard_owner.argtypes = [ci, ci, ci]
    L.achip_shard_slots.restype = ci
    L.achip_shard_slots.argtypes = [ci, ci]
    L.asciichat_hip_grid_create.restype = ci
    L.asciichat_hip_grid_create.argtypes = [C.POINTER(vp), vp, C.POINTER(ci), C.POINTER(ci), C.c_char_p, ci, ci, ci]
    L.asciichat_hip_grid_owner.restype = ci
    L.asciichat_hip_grid_owner.argtypes = [vp, ci]
    L.asciichat_hip_grid_exchange.restype = ci
    L.asciichat_hip_grid_exchange.argtypes = [vp, C.POINTER(vp), vp]
    L.asciichat_hip_grid_composite_dev.restype = vp
    L.asciichat_hip_grid_composite_dev.argtypes = [vp]
    L.asciichat_hip_grid_geometry.restype = C.POINTER(Composite)
    L.asciichat_hip_grid_geometry.argtypes = [vp]
    L.asciichat_hip_grid_set_direct.restype = ci
    L.asciichat_hip_grid_set_direct.argtypes = [vp, ci]
    L.asciichat_hip_grid_destroy.restype = None
    L.asciichat_hip_grid_destroy.argtypes = [vp]
    L.asciichat_hip_plan_render_crc.restype = ci
    L.asciichat_hip_plan_render_crc.argtypes = [vp, vp, sz, vp, vp, vp]
    L.asciichat_hip_plan_render_packets.restype = ci
    L.asciichat_hip_plan_render_packets.argtypes = [vp, vp, sz, vp, vp, vp, vp, vp, vp]
    L.asciichat_hip_plan_render_crc_profiled.restype = ci
    L.asciichat_hip_plan_render_crc_profiled.argtypes = [vp, vp, sz, vp, vp, vp, vp]
    L.asciichat_hip_plan_set_fused_crc.restype = ci
    L.asciichat_hip_plan_set_fused_crc.argtypes = [vp, ci]
    L.asciichat_hip_plan_has_fused_crc.restype = ci
    L.asciichat_hip_plan_has_fused_crc.argtypes = [vp]
    L.asciichat_hip_packets_from_crc.restype = ci
    L.asciichat_hip_packets_from_crc.argtypes = [vp, vp, ci, vp, vp, vp, vp]
    L.asciichat_hip_set_coalesce_min_callers.restype = ci
    L.asciichat_hip_set_coalesce_min_callers.argtypes = [ci]
    L.asciichat_hip_schedule_create.restype = ci
    L.asciichat_hip_schedule_create.argtypes = [C.POINTER(vp), C.POINTER(vp), ci, C.POINTER(vp), C.POINTER(vp), sz, ci, ci, ci]
    L.asciichat_hip_schedule_launch.restype = ci
    L.asciichat_hip_schedule_launch.argtypes = [vp, vp]
    L.asciichat_hip_schedule_destroy.restype = None
    L.asciichat_hip_schedule_destroy.argtypes = [vp]
    L.asciichat_hip_resize.restype = ci
    L.asciichat_hip_resize.argtypes = [vp, ci, ci, vp, ci, ci, vp]
    L.asciichat_hip_composite.restype = ci
    L.asciichat_hip_composite.argtypes = [C.POINTER(Composite), vp, vp]
    L.asciichat_hip_composite_upload.restype = ci
    L.asciichat_hip_composite_upload.argtypes = [C.POINTER(Composite), C.POINTER(vp)]
    L.asciichat_hip_apply_color_filter.restype = ci
    L.asciichat_hip_apply_color_filter.argtypes = [vp, ci, ci, ci, ci, vp]
    L.asciichat_hip_image_flip.restype = ci
    L.asciichat_hip_image_flip.argtypes = [vp, vp, ci, ci, ci, ci, vp]
    L.asciichat_hip_free.restype = None
    L.asciichat_hip_free.argtypes = [vp]
    # achip_host.h
    L.aspect_ratio.restype = None
    L.aspect_ratio.argtypes = [ss, ss, ss, ss, C.c_bool, C.POINTER(ss), C.POINTER(ss)]
    L.achip_mode_from_caps.restype = ci
    L.achip_mode_from_caps.argtypes = [ci, ci]
    L.achip_frame_setup.restype = ci
    L.achip_frame_setup.argtypes = [C.POINTER(Frame), vp, ci, ci, ss, ss, ci, C.c_bool, C.c_bool, C.c_bool]
    L.achip_frame_set_display_ops.restype = ci
    L.achip_frame_set_display_ops.argtypes = [C.POINTER(Frame), C.c_bool, C.c_bool, ci]
    L.achip_rainbow_color.restype = None
    L.achip_rainbow_color.argtypes = [C.c_float, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.POINTER(C.c_uint8)]
    L.achip_frame_set_rainbow.restype = ci
    L.achip_frame_set_rainbow.argtypes = [C.POINTER(Frame), C.c_float]
    L.color_filter_calculate_rainbow.restype = None
    L.color_filter_calculate_rainbow.argtypes = [C.c_float, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.POINTER(C.c_uint8)]
    L.rainbow_replace_ansi_colors.restype = vp
    L.rainbow_replace_ansi_colors.argtypes = [C.c_char_p, C.c_float]
    L.achip_frame_set_dither_style.restype = ci
    L.achip_frame_set_dither_style.argtypes = [C.POINTER(Frame), C.c_bool, C.c_bool]
    L.achip_frame_identity.restype = ci
    L.achip_frame_identity.argtypes = [C.POINTER(Frame), vp, ci, ci]
    L.achip_out_bound.restype = sz
    L.achip_out_bound.argtypes = [ci, C.POINTER(Frame)]
    L.achip_grid_layout.restype = None
    L.achip_grid_layout.argtypes = [C.POINTER(ci), C.POINTER(ci), ci, ci, ci, C.POINTER(ci), C.POINTER(ci)]
    L.achip_composite_setup.restype = None
    L.achip_composite_setup.argtypes = [C.POINTER(Composite), C.POINTER(vp), C.POINTER(ci), C.POINTER(ci), ci, ci, ci]
    # drop-in layer (asciichat_render.h)
    L.ascii_convert.restype = vp
    L.ascii_convert.argtypes = [C.POINTER(Image), ss, ss, C.c_bool, C.c_bool, C.c_bool, C.c_char_p, C.c_char_p]
    L.ascii_convert_with_capabilities_into.restype = ci
    L.ascii_convert_with_capabilities_into.argtypes = [C.POINTER(Image), ss, ss, C.POINTER(TermCaps), C.c_bool, C.c_bool,
                                                       C.c_char_p, vp, sz, C.POINTER(sz)]
    L.ascii_convert_with_capabilities.restype = vp
    L.ascii_convert_with_capabilities.argtypes = [C.POINTER(Image), ss, ss, C.POINTER(TermCaps), C.c_bool, C.c_bool,
                                                  C.c_char_p]
    L.image_print_with_capabilities.restype = vp
    L.image_print_with_capabilities.argtypes = [C.POINTER(Image), C.POINTER(TermCaps), C.c_char_p]
    for name in ("image_print", "image_print_color", "image_print_256color", "image_print_16color",
                 "image_print_color_background"):
        f = getattr(L, name)
        f.restype = vp
        f.argtypes = [C.POINTER(Image), C.c_char_p]
    L.image_print_color_simd.restype = vp
    L.image_print_color_simd.argtypes = [C.POINTER(Image), C.c_bool, C.c_bool, C.c_char_p]
    L.image_print_16color_dithered.restype = vp
    L.image_print_16color_dithered.argtypes = [C.POINTER(Image), C.c_char_p]
    L.image_print_16color_dithered_with_background.restype = vp
    L.image_print_16color_dithered_with_background.argtypes = [C.POINTER(Image), C.c_bool, C.c_char_p]
    L.rgb_to_truecolor_halfblocks_scalar.restype = vp
    L.rgb_to_truecolor_halfblocks_scalar.argtypes = [vp, ci, ci, ci]
    for name in ("rgb_to_256color_halfblocks_scalar", "rgb_to_16color_halfblocks_scalar", "rgb_to_halfblocks_scalar"):
        f = getattr(L, name)
        f.restype = vp
        f.argtypes = [vp, ci, ci, ci, C.c_char_p]
    L.ascii_pad_frame_width.restype = vp
    L.ascii_pad_frame_width.argtypes = [C.c_char_p, sz]
    L.ascii_pad_frame_height.restype = vp
    L.ascii_pad_frame_height.argtypes = [C.c_char_p, sz]
    L.ascii_create_grid.restype = vp
    L.ascii_create_grid.argtypes = [C.POINTER(FrameSource), ci, ci, ci, C.POINTER(sz)]
    L.asciichat_hip_set_option_render_mode.restype = None
    L.asciichat_hip_set_option_render_mode.argtypes = [ci]
    L.image_new.restype = C.POINTER(Image)
    L.image_new.argtypes = [sz, sz]
    L.image_new_from_pool.restype = C.POINTER(Image)
    L.image_new_from_pool.argtypes = [sz, sz]
    L.image_new_copy.restype = C.POINTER(Image)
    L.image_new_copy.argtypes = [C.POINTER(Image)]
    for name in ("image_destroy", "image_destroy_to_pool", "image_clear"):
        f = getattr(L, name)
        f.restype = None
        f.argtypes = [C.POINTER(Image)]
    L.image_resize.restype = None
    L.image_resize.argtypes = [C.POINTER(Image), C.POINTER(Image)]
    L.rgb_to_256color.restype = C.c_uint8
    L.rgb_to_256color.argtypes = [C.c_uint8] * 3
    L.rgb_to_16color.restype = C.c_uint8
    L.rgb_to_16color.argtypes = [C.c_uint8] * 3
    L.rep_is_profitable.restype = C.c_bool
    L.rep_is_profitable.argtypes = [C.c_uint32]
    for name in ("append_truecolor_fg", "append_truecolor_bg"):
        f = getattr(L, name)
        f.restype = vp
        f.argtypes = [vp, C.c_uint8, C.c_uint8, C.c_uint8]
    L.buffer_pool_alloc.restype = vp
    L.buffer_pool_alloc.argtypes = [vp, sz]
    L.buffer_pool_free.restype = None
    L.buffer_pool_free.argtypes = [vp, vp, sz]
    L.buffer_pool_is_pinned.restype = C.c_bool
    L.buffer_pool_is_pinned.argtypes = [vp]
    L.buffer_pool_pinned_blocks.restype = sz
    L.buffer_pool_pinned_blocks.argtypes = [vp]
    L.buffer_pool_get_global.restype = vp
    L.buffer_pool_get_stats.restype = None
    L.buffer_pool_get_stats.argtypes = [vp, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]
    L.buffer_pool_cleanup_global.restype = None
    L.free = C.CDLL(None).free
    L.free.argtypes = [vp]
    return L


def last_error():
    return lib().asciichat_hip_last_error().decode("utf-8", "replace")


def take_string(ptr):
    """Copy a malloc'd NUL-terminated result to bytes and free() it (None for NULL)."""
    if not ptr:
        return None
    out = C.string_at(ptr)
    lib().free(ptr)
    return out


def frame_setup(src_ptr, src_w, src_h, width, height, render_mode, wants_padding=False, use_aspect=False,
                stretch=False):
    f = Frame()
    rc = lib().achip_frame_setup(C.byref(f), src_ptr, src_w, src_h, width, height, render_mode, wants_padding,
                                 use_aspect, stretch)
    return f if rc == 0 else None


class Plan:
    """Batch plan: N device-resident frames -> output slab (asciichat_hip_plan_*)."""

    def __init__(self, mode, palette, frames):
        self.n = len(frames)
        self._arr = (Frame * self.n)(*frames)
        self._h = C.c_void_p()
        p = palette.encode("utf-8") if isinstance(palette, str) else palette
        rc = lib().asciichat_hip_plan_create(C.byref(self._h), mode, p, self._arr, self.n)
        if rc != 0:
            raise RuntimeError(f"asciichat_hip_plan_create failed ({rc}): {last_error()}")
        self.stride = int(lib().asciichat_hip_plan_out_stride(self._h))

    @property
    def variant(self):
        return lib().asciichat_hip_plan_get_variant(self._h)

    def set_variant(self, v):
        rc = lib().asciichat_hip_plan_set_variant(self._h, v)
        if rc != 0:
            raise RuntimeError(f"set_variant({v}) failed: {last_error()}")

    @property
    def parts(self):
        return lib().asciichat_hip_plan_get_parts(self._h)

    @property
    def uniform(self):
        """True when launches pass the batch's common descriptor in the kernel arguments"""
        return bool(lib().asciichat_hip_plan_get_uniform(self._h))

    def set_concurrency(self, launches_in_flight):
        rc = lib().asciichat_hip_plan_set_concurrency(self._h, launches_in_flight)
        if rc != 0:
            raise RuntimeError(f"set_concurrency({launches_in_flight}) failed: {last_error()}")

    def set_uniform(self, allow):
        lib().asciichat_hip_plan_set_uniform(self._h, 1 if allow else 0)

    @property
    def exact_length(self):
        """True when the packed entry points are ONE launch that writes the frames at their exact lengths itself (no slab;
        frames in completion order, off_out tells where)"""
        return bool(lib().asciichat_hip_plan_get_exact_length(self._h))

    @property
    def length_first(self):
        """True when render_packed / render_packets_packed may take the length-first form (frames beyond 48 KB, one launch)"""
        L = lib()
        L.asciichat_hip_plan_get_length_first.restype = C.c_int
        L.asciichat_hip_plan_get_length_first.argtypes = [C.c_void_p]
        return bool(L.asciichat_hip_plan_get_length_first(self._h))

    def set_exact_length(self, mode):
        if lib().asciichat_hip_plan_set_exact_length(self._h, mode) != 0:
            raise RuntimeError(f"set_exact_length({mode}) failed: {last_error()}")

    def set_split(self, rows_per_part):
        rc = lib().asciichat_hip_plan_set_split(self._h, rows_per_part)
        if rc != 0:
            raise RuntimeError(f"set_split({rows_per_part}) failed: {last_error()}")

    def update(self, frames, stream=0):
        self._arr = frames if isinstance(frames, C.Array) else (Frame * self.n)(*frames)
        rc = lib().asciichat_hip_plan_update(self._h, self._arr, stream)
        if rc != 0:
            raise RuntimeError(f"plan_update failed: {last_error()}")
        self.stride = int(lib().asciichat_hip_plan_out_stride(self._h))

    def render(self, out_ptr, out_stride, len_ptr, stream=0, first=0, count=None):
        count = self.n - first if count is None else count
        rc = lib().asciichat_hip_plan_render_range(self._h, first, count, out_ptr, out_stride, len_ptr, stream)
        if rc != 0:
            raise RuntimeError(f"plan_render failed ({rc}): {last_error()}")

    def render_packed(self, slab_ptr, out_stride, len_ptr, dst_ptr, dst_capacity, off_ptr=None, len_out_ptr=None, stream=0):
        """render + compaction (asciichat_hip_plan_render_packed): frame i -> dst + off[i], off[n] = total bytes"""
        rc = lib().asciichat_hip_plan_render_packed(self._h, slab_ptr, out_stride, len_ptr, dst_ptr, dst_capacity, off_ptr,
                                                    len_out_ptr, stream)
        if rc != 0:
            raise RuntimeError(f"plan_render_packed failed ({rc}): {last_error()}")

    def render_crc(self, out_ptr, out_stride, len_ptr, crc_ptr, stream=0):
        rc = lib().asciichat_hip_plan_render_crc(self._h, out_ptr, out_stride, len_ptr, crc_ptr, stream)
        if rc != 0:
            raise RuntimeError(f"plan_render_crc failed ({rc}): {last_error()}")

    def render_packets(self, out_ptr, out_stride, len_ptr, dims_ptr, crc_ptr, hdr_ptr, pkt_ptr, stream=0):
        rc = lib().asciichat_hip_plan_render_packets(self._h, out_ptr, out_stride, len_ptr, dims_ptr, crc_ptr, hdr_ptr,
                                                     pkt_ptr, stream)
        if rc != 0:
            raise RuntimeError(f"plan_render_packets failed ({rc}): {last_error()}")

    def render_packets_packed(self, slab_ptr, out_stride, len_ptr, dims_ptr, crc_ptr, hdr_ptr, pkt_ptr, dst_ptr, dst_capacity,
                              off_ptr=None, len_out_ptr=None, stream=0):
        """render + wire stage + compaction (asciichat_hip_plan_render_packets_packed)"""
        rc = lib().asciichat_hip_plan_render_packets_packed(self._h, slab_ptr, out_stride, len_ptr, dims_ptr, crc_ptr, hdr_ptr,
                                                            pkt_ptr, dst_ptr, dst_capacity, off_ptr, len_out_ptr, stream)
        if rc != 0:
            raise RuntimeError(f"plan_render_packets_packed failed ({rc}): {last_error()}")

    def set_fused_crc(self, mode):
        """-1 automatic (fused where it is the faster form), 0 never, 1 wherever the geometry carries it"""
        rc = lib().asciichat_hip_plan_set_fused_crc(self._h, mode)
        if rc == 30:  # ASCIICHAT_HIP_ERR_NOT_SUPPORTED: the setting is kept, this geometry / build has no fused form
            return False
        if rc != 0:
            raise RuntimeError(f"set_fused_crc({mode}) failed: {last_error()}")
        return True

    @property
    def fused_crc(self):
        return bool(lib().asciichat_hip_plan_has_fused_crc(self._h))

    def render_profiled(self, out_ptr, out_stride, len_ptr, prof_ptr, stream=0):
        rc = lib().asciichat_hip_plan_render_profiled(self._h, out_ptr, out_stride, len_ptr, prof_ptr, stream)
        if rc != 0:
            raise RuntimeError(f"plan_render_profiled failed ({rc}): {last_error()}")

    def close(self):
        if self._h:
            lib().asciichat_hip_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Schedule:
    """A round-robin tick loop issued from C (asciichat_hip_render_many): step k renders plans[k % P] on
    streams[k % S] into outs[k % S].  One FFI call per K steps instead of one per launch."""

    def __init__(self, plans, out_ptrs, len_ptrs, out_stride, stream_handles):
        P, S = len(plans), len(stream_handles)
        assert P % S == 0 and len(out_ptrs) == S and len(len_ptrs) == S
        self._plans = (C.c_void_p * P)(*[p._h for p in plans])
        self._outs = (C.c_void_p * S)(*out_ptrs)
        self._lens = (C.c_void_p * S)(*len_ptrs)
        self._streams = (C.c_void_p * S)(*stream_handles)
        self.P, self.S, self.stride = P, S, out_stride

    def issue(self, first_step, n_steps):
        rc = lib().asciichat_hip_render_many(self._plans, self.P, self._outs, self._lens, self.stride, self._streams,
                                             self.S, first_step, n_steps)
        if rc != 0:
            raise RuntimeError(f"render_many failed ({rc}): {last_error()}")

    def issue_profiled(self, first_step, n_steps, prof_ptr, prof_stride_words):
        rc = lib().asciichat_hip_render_many_profiled(self._plans, self.P, self._outs, self._lens, self.stride,
                                                      self._streams, self.S, first_step, n_steps, prof_ptr,
                                                      prof_stride_words)
        if rc != 0:
            raise RuntimeError(f"render_many_profiled failed ({rc}): {last_error()}")

    def wait(self):
        rc = lib().asciichat_hip_streams_wait(self._streams, self.S)
        if rc != 0:
            raise RuntimeError(f"streams_wait failed ({rc}): {last_error()}")

    def graph(self, first_step, n_steps):
        """The same n_steps captured into a HIP graph (asciichat_hip_schedule_create); cached per (first % P, n)."""
        key = (first_step % self.P, n_steps)
        cache = self.__dict__.setdefault("_graphs", {})
        if key not in cache:
            h = C.c_void_p()
            rc = lib().asciichat_hip_schedule_create(C.byref(h), self._plans, self.P, self._outs, self._lens, self.stride,
                                                     self.S, key[0], n_steps)
            if rc != 0:
                raise RuntimeError(f"schedule_create failed ({rc}): {last_error()}")
            cache[key] = h
        return cache[key]

    def replay(self, first_step, n_steps, stream):
        rc = lib().asciichat_hip_schedule_launch(self.graph(first_step, n_steps), stream)
        if rc != 0:
            raise RuntimeError(f"schedule_launch failed ({rc}): {last_error()}")

    def close(self):
        for h in self.__dict__.pop("_graphs", {}).values():
            lib().asciichat_hip_schedule_destroy(h)


def pack_frames(slab_ptr, stride, len_ptr, n, dst_ptr, dst_capacity, off_ptr=None, len_out_ptr=None, stream=0):
    rc = lib().asciichat_hip_pack_frames(slab_ptr, stride, len_ptr, n, dst_ptr, dst_capacity, off_ptr, len_out_ptr, stream)
    if rc != 0:
        raise RuntimeError(f"pack_frames failed ({rc}): {last_error()}")


class HostBuffer:
    """Mapped pinned host memory (asciichat_hip_host_alloc): .host for the CPU, .dev for kernels."""

    def __init__(self, nbytes):
        h, d = C.c_void_p(), C.c_void_p()
        rc = lib().asciichat_hip_host_alloc(nbytes, C.byref(h), C.byref(d))
        if rc != 0:
            raise RuntimeError(f"host_alloc failed ({rc}): {last_error()}")
        self.host, self.dev, self.nbytes = h.value, d.value, nbytes

    def view(self, offset=0, nbytes=None):
        """numpy uint8 view of the host side"""
        import numpy as np
        n = self.nbytes - offset if nbytes is None else nbytes
        return np.ctypeslib.as_array((C.c_uint8 * n).from_address(self.host + offset))

    def close(self):
        if self.host:
            lib().asciichat_hip_host_free(self.host)
            self.host = self.dev = None


COMM_ID_BYTES = 128


def comm_unique_id():
    """128-byte RCCL unique id (rank 0 creates it, every rank passes it to Comm)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = lib().asciichat_hip_comm_unique_id(buf, COMM_ID_BYTES)
    if rc != 0:
        raise RuntimeError(f"comm_unique_id failed ({rc}): {last_error()}")
    return buf.raw


class Comm:
    """RCCL communicator of the C-ABI (asciichat_hip_comm_*): one process per GPU."""

    def __init__(self, world, rank, unique_id):
        self._h = C.c_void_p()
        rc = lib().asciichat_hip_comm_init(C.byref(self._h), world, rank, unique_id, len(unique_id))
        if rc != 0:
            raise RuntimeError(f"comm_init failed ({rc}): {last_error()}")
        self.world, self.rank = world, rank

    def all_gather(self, send_ptr, recv_ptr, bytes_per_rank, stream=0):
        rc = lib().asciichat_hip_comm_all_gather(self._h, send_ptr, recv_ptr, bytes_per_rank, stream)
        if rc != 0:
            raise RuntimeError(f"comm_all_gather failed ({rc}): {last_error()}")

    @property
    def count(self):
        """ranks the communicator itself reports (ncclCommCount)"""
        return lib().asciichat_hip_comm_count(self._h)

    def all_gather_packed(self, slab_ptr, stride, len_ptr, slots_per_rank, packed_ptr, capacity_per_rank, stream=0):
        """-> (offsets[world*slots], lengths[world*slots], bytes every rank contributed)"""
        n = self.world * slots_per_rank
        off, ln, blk = (C.c_uint64 * n)(), (C.c_uint32 * n)(), C.c_size_t()
        rc = lib().asciichat_hip_comm_all_gather_packed(self._h, slab_ptr, stride, len_ptr, slots_per_rank, packed_ptr,
                                                        capacity_per_rank, off, ln, C.byref(blk), stream)
        if rc != 0:
            raise RuntimeError(f"comm_all_gather_packed failed ({rc}): {last_error()}")
        return list(off), list(ln), blk.value

    def all_gather_frames(self, slab_ptr, stride, len_ptr, slots_per_rank, packed_ptr, capacity_per_rank, form=-1, stream=0):
        """Both gathers behind one entry (form 0 packed, 1 slab, -1: ASCIICHAT_HIP_GATHER) ->
        (base pointer, offsets[world*slots], lengths[world*slots] or None, bytes every rank contributed, form taken)"""
        L = lib()
        L.asciichat_hip_comm_all_gather_frames.restype = C.c_int
        L.asciichat_hip_comm_all_gather_frames.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p,
                                                           C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32),
                                                           C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.c_void_p]
        n = self.world * slots_per_rank
        off, ln, blk, base, took = (C.c_uint64 * n)(), (C.c_uint32 * n)(), C.c_size_t(), C.c_void_p(), C.c_int()
        rc = L.asciichat_hip_comm_all_gather_frames(self._h, form, slab_ptr, stride, len_ptr, slots_per_rank, packed_ptr, capacity_per_rank,
                                                    C.byref(base), off, ln, C.byref(blk), C.byref(took), stream)
        if rc != 0:
            raise RuntimeError(f"comm_all_gather_frames failed ({rc}): {last_error()}")
        return base.value, list(off), (list(ln) if took.value == 0 else None), blk.value, took.value

    def all_gather_slab(self, slab_ptr, stride, len_ptr, slots_per_rank, stream=0):
        rc = lib().asciichat_hip_comm_all_gather_slab(self._h, slab_ptr, stride, len_ptr, slots_per_rank, stream)
        if rc != 0:
            raise RuntimeError(f"comm_all_gather_slab failed ({rc}): {last_error()}")

    def close(self):
        if self._h:
            lib().asciichat_hip_comm_destroy(self._h)
            self._h = C.c_void_p()


class Grid:
    """The pixel-space grid across GPUs (asciichat_hip_grid_*): exchange() resizes the sources this rank owns into
    their tiles and all-gathers the tiles; frames whose .comp = composite_dev then render the grid."""

    def __init__(self, comm, src_dims, term_w, term_h, has_video=None):
        n = len(src_dims)
        self._h = C.c_void_p()
        ws = (C.c_int * n)(*[d[0] for d in src_dims])
        hs = (C.c_int * n)(*[d[1] for d in src_dims])
        hv = bytes(1 if v else 0 for v in has_video) if has_video is not None else None
        rc = lib().asciichat_hip_grid_create(C.byref(self._h), comm._h if comm else None, ws, hs, hv, n, term_w, term_h)
        if rc != 0:
            raise RuntimeError(f"grid_create failed ({rc}): {last_error()}")
        self.n = n
        self.composite_dev = lib().asciichat_hip_grid_composite_dev(self._h)
        self.geometry = lib().asciichat_hip_grid_geometry(self._h).contents

    def owner(self, source):
        return lib().asciichat_hip_grid_owner(self._h, source)

    def set_direct(self, on):
        """one GPU: render straight from the sources (no tiles, no resize, no collective); refreshes composite_dev"""
        rc = lib().asciichat_hip_grid_set_direct(self._h, 1 if on else 0)
        if rc != 0:
            raise RuntimeError(f"grid_set_direct failed ({rc}): {last_error()}")
        self.composite_dev = lib().asciichat_hip_grid_composite_dev(self._h)

    def exchange(self, local_ptrs, stream=0):
        """local_ptrs: {source index: device pointer} for the sources this rank owns"""
        arr = (C.c_void_p * self.n)(*[local_ptrs.get(k) for k in range(self.n)])
        rc = lib().asciichat_hip_grid_exchange(self._h, arr, stream)
        if rc != 0:
            raise RuntimeError(f"grid_exchange failed ({rc}): {last_error()}")

    def close(self):
        if self._h:
            lib().asciichat_hip_grid_destroy(self._h)
            self._h = C.c_void_p()


class FrameTable:
    """Device-resident latest-frame table (asciichat_hip_frame_table_*)."""

    def __init__(self, n_slots):
        self._h = C.c_void_p()
        if lib().asciichat_hip_frame_table_create(C.byref(self._h), n_slots) != 0:
            raise RuntimeError(f"frame_table_create failed: {last_error()}")

    def publish(self, slot, blob, stream=0):
        rc = lib().asciichat_hip_frame_table_publish(self._h, slot, bytes(blob), len(blob), stream)
        if rc != 0:
            raise RuntimeError(f"frame_table_publish failed: {last_error()}")

    def publish_rows(self, slot, blob, targets, stream=0):
        """publish only the rows that renders described by `targets` (Frame descriptors) will sample; blob: bytes, or
        (address, size) of a host buffer"""
        arr = (Frame * len(targets))(*targets)
        if isinstance(blob, tuple):
            rc = lib().asciichat_hip_frame_table_publish_rows(self._h, slot, blob[0], blob[1], arr, len(targets), stream)
        else:
            b = bytes(blob)
            rc = lib().asciichat_hip_frame_table_publish_rows(self._h, slot, b, len(b), arr, len(targets), stream)
        if rc != 0:
            raise RuntimeError(f"frame_table_publish_rows failed: {last_error()}")

    def publish_rows_batch(self, slots, blobs, targets, stream=0):
        """one call for a whole tick: blobs = [(address, size)] of host buffers (or a prepared (c_void_p array, c_size_t
        array) pair), slots = their table slots (a list or a prepared c_int array)"""
        n = len(slots)
        arr = targets if isinstance(targets, C.Array) else (Frame * len(targets))(*targets)
        sl = slots if isinstance(slots, C.Array) else (C.c_int * n)(*slots)
        if isinstance(blobs, tuple) and isinstance(blobs[0], C.Array):
            ptrs, sizes = blobs
        else:
            ptrs, sizes = (C.c_void_p * n)(*[b[0] for b in blobs]), (C.c_size_t * n)(*[b[1] for b in blobs])
        rc = lib().asciichat_hip_frame_table_publish_rows_batch(self._h, sl, ptrs, sizes, n, arr, len(arr), stream)
        if rc != 0:
            raise RuntimeError(f"frame_table_publish_rows_batch failed: {last_error()}")

    def stage(self, slot, blob, target):
        """gather what `target` samples of the blob ((address, size) or bytes) into the tick's pinned block; any thread"""
        t = target if isinstance(target, C.Array) else (Frame * 1)(target)
        if isinstance(blob, tuple):
            rc = lib().asciichat_hip_frame_table_stage(self._h, slot, blob[0], blob[1], t)
        else:
            b = bytes(blob)
            rc = lib().asciichat_hip_frame_table_stage(self._h, slot, C.cast(C.c_char_p(b), C.c_void_p), len(b), t)
        if rc != 0:
            raise RuntimeError(f"frame_table_stage failed: {last_error()}")

    def commit(self, stream=0):
        """one DMA for everything staged since the last commit"""
        if lib().asciichat_hip_frame_table_commit(self._h, stream) != 0:
            raise RuntimeError(f"frame_table_commit failed: {last_error()}")

    def publish_sampled_batch(self, slots, blobs, targets, stream=0):
        """a whole tick: stage every blob (on the library's ingest threads) + commit; targets: one Frame for all, or one per blob"""
        n = len(slots)
        arr = targets if isinstance(targets, C.Array) else (Frame * len(targets))(*targets)
        sl = slots if isinstance(slots, C.Array) else (C.c_int * n)(*slots)
        if isinstance(blobs, tuple) and isinstance(blobs[0], C.Array):
            ptrs, sizes = blobs
        else:
            ptrs, sizes = (C.c_void_p * n)(*[b[0] for b in blobs]), (C.c_size_t * n)(*[b[1] for b in blobs])
        rc = lib().asciichat_hip_frame_table_publish_sampled_batch(self._h, sl, ptrs, sizes, n, arr, len(arr), stream)
        if rc != 0:
            raise RuntimeError(f"frame_table_publish_sampled_batch failed: {last_error()}")

    def latest_frames(self, slots, frames, stream=0):
        """frames[i].src = the latest device frame of slots[i] (None when it has none of that descriptor's geometry);
        slots: c_int array, frames: (Frame * n) array updated in place.  -> number of descriptors with a source"""
        n = len(slots)
        sl = slots if isinstance(slots, C.Array) else (C.c_int * n)(*slots)
        rc = lib().asciichat_hip_frame_table_latest_frames(self._h, sl, n, stream, frames)
        if rc < 0:
            raise RuntimeError(f"frame_table_latest_frames failed: {last_error()}")
        return rc

    def publish_at(self, slot, address, size, stream=0):
        """publish a blob that already sits in host memory at `address` (e.g. a block of the pinned pool)"""
        rc = lib().asciichat_hip_frame_table_publish(self._h, slot, C.c_char_p(address), size, stream)
        if rc != 0:
            raise RuntimeError(f"frame_table_publish failed: {last_error()}")

    def latest(self, slot, stream=0):
        """-> (device pointer or None, width, height, generation)"""
        p, w, h, g = C.c_void_p(), C.c_int(), C.c_int(), C.c_uint64()
        rc = lib().asciichat_hip_frame_table_latest(self._h, slot, stream, C.byref(p), C.byref(w), C.byref(h), C.byref(g))
        if rc != 0:
            raise RuntimeError(f"frame_table_latest failed: {last_error()}")
        return p.value, w.value, h.value, g.value

    def forget_stream(self, stream):
        lib().asciichat_hip_frame_table_forget_stream(self._h, stream)

    def close(self):
        if self._h:
            lib().asciichat_hip_frame_table_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
