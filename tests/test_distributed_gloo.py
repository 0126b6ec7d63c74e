"""world_size-2 gloo test of the multi-GPU path (sharding + all-gathers) on CPU.  Compute is the kernel
emulator (the same kernel source the GPU runs); on the GPU box bench.py/the driver run this path over RCCL."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _worker(rank, world, port, q):
    try:
        import torch
        import torch.distributed as dist

        import emu
        import orc
        from __graft_entry__ import load_package

        pkg = load_package()
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        EL = emu.lib()

        # ---- (1) a logical batch of 5 independent frames sharded over 2 ranks, outputs all-gathered ----
        imgs = [orc.frame_hash_noise(96, 54, 100 + i) for i in range(5)]
        frames = [emu.frame_for_convert(im, 40, 12, 0) for im in imgs]
        arr = (emu.Frame * 5)(*frames)
        mode = 1
        lut = emu.make_lut(orc.PALETTE_STANDARD)
        stride = max(int(EL.achip_out_bound(mode, C.byref(arr[i]))) for i in range(5))
        stride = (stride + 1 + 15) // 16 * 16 + 16
        sb = pkg.distributed.ShardedBatch(torch, 5, stride, world, rank, "cpu")
        assert (sb.per, sb.first, sb.count) == ((3, 0, 3) if rank == 0 else (3, 3, 2))

        def render_range(first, count, out_ptr, len_ptr):
            sub = (emu.Frame * count)(*[arr[first + i] for i in range(count)])
            base = (out_ptr + 15) // 16 * 16
            assert base == out_ptr  # torch CPU allocations are 64-byte aligned
            assert EL.emu_render_batch(mode, 2, sub, count, C.byref(lut), out_ptr, stride, len_ptr) == 0

        sb.render_local(render_range)
        sb.all_gather(dist)
        for i in range(5):
            assert sb.frame_bytes(i) == orc.convert_with_caps(imgs[i], 40, 12, 3, 0), (rank, i)

        # ---- (2) K4: 9 sources dealt over 2 ranks, tiles all-gathered, every rank renders the composite ----
        srcs = [orc.frame_hash_noise(192, 108, 10 + i) if i % 2 else orc.frame_bars(192, 108, i) for i in range(9)]
        per, first, count = pkg.distributed.shard_bounds(9, world, rank)
        local = {k: torch.from_numpy(srcs[k]) for k in range(first, first + count)}
        ptrs = (C.c_void_p * 9)(*[1] * 9)  # geometry only: every rank knows all dims, not all pixels
        ws = (C.c_int * 9)(*[192] * 9)
        hs = (C.c_int * 9)(*[108] * 9)
        comp = emu.Composite()
        EL.achip_composite_setup(C.byref(comp), ptrs, ws, hs, 9, 160, 48)

        class Backend:
            def resize(self, src, sw, sh, dst, dw, dh):
                EL.emu_resize_nn(src, sw, sh, dst, dw, dh, EL.achip_nn_ratio(sw, dw), EL.achip_nn_ratio(sh, dh))

            def sync(self):
                pass

        tiles, tstride, comp2 = pkg.distributed.gather_grid_tiles(torch, dist, Backend(), comp, local, world, rank, "cpu")
        ref = orc.composite(srcs, 160, 48)
        for mode, (cl, rm) in ((1, (3, 0)), (5, (3, 2))):
            h = 96 if rm == 2 else 48
            f = emu.Frame()
            assert EL.achip_frame_setup(C.byref(f), None, 160, 96, 160, h, rm, True, True, False) == 0
            f.comp = C.addressof(comp2)
            got = emu.render_frames(mode, [f], orc.PALETTE_STANDARD, 2)[0]
            assert got == orc.convert_with_caps(ref, 160, h, cl, rm, True, True, False), (rank, mode)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback

        q.put((rank, "FAIL: " + traceback.format_exc()))


def test_sharded_batch_and_grid_tiles_world2():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=240) for _ in procs]
    [p.join(30) for p in procs]
    assert all(r[1] == "ok" for r in res), res


def test_shard_bounds():
    from __graft_entry__ import load_package

    d = load_package().distributed
    assert [d.shard_bounds(256, 8, r)[1:] for r in range(8)] == [(32 * r, 32) for r in range(8)]
    # balanced: nine sources over eight GPUs leave no rank idle (the same rule as achip_shard_bounds in comm.c)
    assert [d.shard_bounds(9, 8, r)[1:] for r in range(8)] == [(0, 2), (2, 1), (3, 1), (4, 1), (5, 1), (6, 1), (7, 1), (8, 1)]
    assert d.shard_bounds(5, 2, 1) == (3, 3, 2) and d.shard_bounds(9, 8, 3)[0] == 2
    assert [d.shard_owner(9, 8, k) for k in range(9)] == [0, 0, 1, 2, 3, 4, 5, 6, 7]
