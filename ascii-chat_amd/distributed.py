"""Multi-GPU plumbing for the render path: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in the CPU tests).

Frames are independent units: a logical batch is cut into contiguous per-rank blocks and
each rank renders only its block -- no halo, no reduction, glyph tables replicated (1.3 KB).  The only
collectives on the data path are all-gathers, used where a consumer needs data produced on other ranks:

  * ShardedBatch.all_gather(): fixed-stride output slab + uint32 lengths of every rank's block;
  * gather_grid_tiles(): the <= 9 resized source tiles of a pixel-space grid composite
    (create_multi_source_composite, src/server/stream.c:664-779) so that every rank can render the
    composite for the target clients it owns (BASELINE config 4).

Messages are KB..MB, i.e. latency-bound on xGMI: one collective per batch, never per frame.
Compute is injected through a small backend object so that the same code runs the real library on the
GPU box and the kernel emulator in the CPU tests.
"""
import ctypes as C


def shard_bounds(n_items, world, rank):
    """Contiguous, balanced partition (achip_shard_bounds in comm.c): (slots every rank reserves, first, count) for
    `rank` -- the first n % world ranks hold one item more, so nine sources over eight GPUs leave no rank idle."""
    base, extra = divmod(max(0, n_items), max(1, world))
    first = rank * base + min(rank, extra)
    count = base + (1 if rank < extra else 0)
    slots = (n_items + world - 1) // world if n_items > 0 else 0
    return slots, first, count


def shard_owner(n_items, world, item):
    for r in range(world):
        _, first, count = shard_bounds(n_items, world, r)
        if first <= item < first + count:
            return r
    return -1


class ShardedBatch:
    """A logical batch of n_frames rendered by `world` ranks into one fixed-stride slab layout."""

    def __init__(self, torch, n_frames, stride, world, rank, device):
        self.torch = torch
        self.n, self.stride, self.world, self.rank = n_frames, stride, world, rank
        self.per, self.first, self.count = shard_bounds(n_frames, world, rank)
        # room for world*per frames so that every rank contributes an equal-sized block to the all-gather
        self.slab = torch.zeros(world * self.per * stride, dtype=torch.uint8, device=device)
        self.lens = torch.zeros(world * self.per, dtype=torch.int32, device=device)

    def local_views(self):
        a, b = self.rank * self.per, (self.rank + 1) * self.per
        return self.slab[a * self.stride:b * self.stride], self.lens[a:b]

    def render_local(self, render_range):
        """render_range(first, count, out_ptr, len_ptr): renders this rank's block in place."""
        slab, lens = self.local_views()
        if self.count:
            render_range(self.first, self.count, slab.data_ptr(), lens.data_ptr())

    def all_gather(self, dist, group=None):
        """Every rank ends up with all ranks' frames (one all-gather for bytes, one for lengths)."""
        if self.world == 1:
            return
        slab, lens = self.local_views()
        dist.all_gather_into_tensor(self.slab, slab.clone(), group=group)
        dist.all_gather_into_tensor(self.lens, lens.clone(), group=group)

    def slot_of(self, i):
        """Slab slot of logical frame i: its owner's block starts at owner * per."""
        owner = shard_owner(self.n, self.world, i)
        return owner * self.per + (i - shard_bounds(self.n, self.world, owner)[1])

    def frame_bytes(self, i):
        """Host copy of logical frame i (after all_gather, or local frames before)."""
        k = self.slot_of(i)
        n = int(self.lens[k].item()) & 0xFFFFFFFF
        return bytes(self.slab[k * self.stride:k * self.stride + n].cpu().numpy())


def gather_grid_tiles(torch, dist, backend, comp, local_sources, world, rank, device):
    """Resize this rank's sources into their composite tiles, all-gather the tiles, and return
    (tiles tensor, per-slot byte stride, composite descriptor rewired to read the gathered tiles).

    comp            achip_composite_t filled by achip_composite_setup() with the geometry of ALL sources
                    (every rank knows every source's dimensions; pointers may be dummies)
    local_sources   {source index: uint8 tensor HxWx3 on `device`} for the sources this rank owns (shard_bounds)
    """
    n = comp.n_src
    per, first, count = shard_bounds(n, world, rank)
    tile_stride = 16
    for k in range(n):
        tile_stride = max(tile_stride, (comp.s[k].tile_w * comp.s[k].tile_h * 3 + 15) // 16 * 16)
    tiles = torch.zeros(world * per * tile_stride, dtype=torch.uint8, device=device)
    def slot_of(k):
        owner = shard_owner(n, world, k)
        return owner * per + (k - shard_bounds(n, world, owner)[1])

    for k in range(first, first + count):
        s = comp.s[k]
        if not s.src:
            continue
        src = local_sources[k]
        backend.resize(src.data_ptr(), s.src_w, s.src_h, tiles.data_ptr() + slot_of(k) * tile_stride, s.tile_w, s.tile_h)
    backend.sync()
    if world > 1:
        mine = tiles[rank * per * tile_stride:(rank + 1) * per * tile_stride].clone()
        dist.all_gather_into_tensor(tiles, mine)
    # the composite now samples the already-resized tiles: identity ratio, tile-sized "sources"
    out = type(comp)()
    C.memmove(C.byref(out), C.byref(comp), C.sizeof(comp))
    for k in range(n):
        s = out.s[k]
        if not s.src:
            continue
        s.src = tiles.data_ptr() + slot_of(k) * tile_stride
        s.src_w, s.src_h, s.src_stride = s.tile_w, s.tile_h, 3 * s.tile_w
        s.x_ratio = s.y_ratio = 65537  # ((n << 16) / n) + 1
    return tiles, tile_stride, out


class GpuBackend:
    """Compute backend for gather_grid_tiles() on the GPU box: the library's resize kernel on torch's stream."""

    def __init__(self, torch, lib):
        self.torch, self.lib = torch, lib

    def resize(self, src_ptr, sw, sh, dst_ptr, dw, dh):
        rc = self.lib.asciichat_hip_resize(src_ptr, sw, sh, dst_ptr, dw, dh,
                                           self.torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError("asciichat_hip_resize failed: %d" % rc)

    def sync(self):
        self.torch.cuda.synchronize()


def render_grid_for_targets(torch, dist, pkg, sources, src_dims, term_w, term_h, targets, palette, world, rank,
                            device="cuda"):
    """BASELINE config 4 end to end: `sources` = {slot: uint8 HxWx3 device tensor} owned by this rank (contiguous
    blocks of slots per rank), `src_dims` = [(w, h)] of ALL sources, `targets` = [(color_level, render_mode,
    wants_padding)] of the target clients this rank renders.  Returns a list of bytes, one frame per target."""
    n = len(src_dims)
    ptrs = (C.c_void_p * n)(*[1] * n)
    ws = (C.c_int * n)(*[d[0] for d in src_dims])
    hs = (C.c_int * n)(*[d[1] for d in src_dims])
    comp = pkg.Composite()
    pkg.lib().achip_composite_setup(C.byref(comp), ptrs, ws, hs, n, term_w, term_h)
    tiles, _, comp2 = gather_grid_tiles(torch, dist, GpuBackend(torch, pkg.lib()), comp, sources, world, rank, device)
    comp_dev = C.c_void_p()
    if pkg.lib().asciichat_hip_composite_upload(C.byref(comp2), C.byref(comp_dev)) != 0:
        raise RuntimeError("composite upload failed")
    out = []
    try:
        for (cl, rm, pad) in targets:
            mode = pkg.lib().achip_mode_from_caps(cl, rm)
            h = term_h * 2 if rm == 2 else term_h  # convert_composite_to_ascii, src/server/stream.c:831
            f = pkg.frame_setup(None, comp.canvas_w, comp.canvas_h, term_w, h, rm, pad, True, False)
            f.comp = comp_dev.value
            plan = pkg.Plan(mode, palette, [f])
            slab = torch.zeros(plan.stride, dtype=torch.uint8, device=device)
            ln = torch.zeros(1, dtype=torch.int32, device=device)
            plan.render(slab.data_ptr(), plan.stride, ln.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            out.append(bytes(slab[:int(ln[0].item()) & 0xFFFFFFFF].cpu().numpy()))
            plan.close()
    finally:
        pkg.lib().asciichat_hip_free(comp_dev)
    del tiles
    return out
