"""One rank of tests/test_comm_two_ranks.py: drives ascii-chat_amd/csrc/comm.c with a world of `world` ranks.

usage: comm_worker.py <rank> <world> <uid_file> <shared_gpu 0|1> [full]

"full" (the world-8 test): the batch is BASELINE's 256 frames (32 per rank at world 8), and frames of configs[4]'s size
(4K -> 400x120 half blocks: 1.8 MB each) go through the packed gather too.

Every rank renders ITS block of a sharded batch, then checks after
  * asciichat_hip_comm_all_gather_slab      -- every frame of every rank against the oracle,
  * asciichat_hip_comm_all_gather_packed    -- the same frames at their packed offsets, and that fewer bytes moved,
  * asciichat_hip_grid_exchange             -- the 3x3 (and, with a client without video, 2x3) grid rendered from the
                                               gathered tiles against the oracle's composite + convert, for source counts
                                               that shard unevenly over the ranks (grid_slot_of).
Prints "ok rank <r>" as its last line; any mismatch raises.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, uid_file, shared = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    import numpy as np
    import torch

    import orc
    from __graft_entry__ import load_package

    pkg = load_package()
    torch.cuda.set_device(0 if shared else rank)
    if rank == 0:
        uid = pkg.comm_unique_id()
        with open(uid_file + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(uid_file + ".tmp", uid_file)
    else:
        t0 = time.time()
        while not os.path.exists(uid_file):
            if time.time() - t0 > 120:
                raise SystemExit("rank 0 never published the unique id")
            time.sleep(0.05)
        uid = open(uid_file, "rb").read()
    comm = pkg.Comm(world, rank, uid)
    assert comm.count == world, f"ncclCommCount says {comm.count}, expected {world}"
    st = torch.cuda.current_stream().cuda_stream
    L = pkg.lib()
    import ctypes as C

    full = len(sys.argv) > 5 and sys.argv[5] == "full"
    # ---- (1) a batch of 7 frames sharded (4, 3): slots = 4, the last slot of rank 1 stays empty ------------------------
    # (full: 256 frames, 32 x 8 at world 8)
    n, W, H = (256 if full else 7), 40, 12
    imgs = [orc.frame_hash_noise(96, 54, 500 + i) if i % 3 else orc.frame_bars(96, 54, i) for i in range(n)]
    slots = L.achip_shard_slots(n, world)
    first, count = C.c_int(), C.c_int()
    L.achip_shard_bounds(n, world, rank, C.byref(first), C.byref(count))
    first, count = first.value, count.value
    dev = [torch.from_numpy(imgs[first + i]).cuda() for i in range(count)]
    frames = [pkg.frame_setup(d.data_ptr(), 96, 54, W, H, 0, False, False, False) for d in dev]
    plan = pkg.Plan(pkg.MODE_TRUE_FG, orc.PALETTE_STANDARD, frames)
    stride = plan.stride
    slab = torch.zeros(world * slots * stride, dtype=torch.uint8, device="cuda")
    ln = torch.full((world * slots,), 0, dtype=torch.int32, device="cuda")
    plan.render(slab.data_ptr() + rank * slots * stride, stride, ln.data_ptr() + 4 * rank * slots, st)
    slab2, ln2 = slab.clone(), ln.clone()
    comm.all_gather_slab(slab.data_ptr(), stride, ln.data_ptr(), slots, st)
    torch.cuda.synchronize()
    exp = [orc.convert_with_caps(im, W, H, 3, 0, False, False, False) for im in imgs]
    host, lens = slab.cpu().numpy(), ln.cpu().numpy().astype("uint32")
    for r in range(world):
        f_r, c_r = C.c_int(), C.c_int()
        L.achip_shard_bounds(n, world, r, C.byref(f_r), C.byref(c_r))
        for i in range(c_r.value):
            s = r * slots + i
            got = host[s * stride:s * stride + int(lens[s])].tobytes()
            assert got == exp[f_r.value + i], f"rank {rank}: slab frame {f_r.value + i} (slot {s}) differs"
    # ---- (2) the same exchange, compacted: lengths first, then max-over-ranks packed bytes ---------------------------------
    packed = torch.zeros(world * slots * stride, dtype=torch.uint8, device="cuda")
    off, plen, blk = comm.all_gather_packed(slab2.data_ptr(), stride, ln2.data_ptr(), slots, packed.data_ptr(), slots * stride, st)
    torch.cuda.synchronize()
    ph = packed.cpu().numpy()
    assert blk % 16 == 0 and blk < slots * stride, (blk, slots * stride)
    for r in range(world):
        f_r, c_r = C.c_int(), C.c_int()
        L.achip_shard_bounds(n, world, r, C.byref(f_r), C.byref(c_r))
        for i in range(c_r.value):
            s = r * slots + i
            assert off[s] % 16 == 0 and r * blk <= off[s] and off[s] + plen[s] <= (r + 1) * blk
            assert ph[off[s]:off[s] + plen[s]].tobytes() == exp[f_r.value + i], f"rank {rank}: packed frame {f_r.value + i} differs"
    # ---- (2a) both forms behind the one entry: by argument and by the environment (what the first A/B on a real node runs)
    for form, env in ((0, None), (1, None), (-1, "slab"), (-1, "packed"), (-1, None)):
        if env is None:
            os.environ.pop("ASCIICHAT_HIP_GATHER", None)
        else:
            os.environ["ASCIICHAT_HIP_GATHER"] = env
        slab3, ln3 = slab2.clone(), ln2.clone()
        packed3 = torch.zeros(world * slots * stride, dtype=torch.uint8, device="cuda")
        base, off3, len3, blk3, took = comm.all_gather_frames(slab3.data_ptr(), stride, ln3.data_ptr(), slots, packed3.data_ptr(),
                                                             slots * stride, form, st)
        torch.cuda.synchronize()
        assert took == (form if form >= 0 else (1 if env == "slab" else 0)), (form, env, took)
        where = packed3 if took == 0 else slab3
        assert base == where.data_ptr() and blk3 == (blk if took == 0 else slots * stride)
        wh, l3 = where.cpu().numpy(), ln3.cpu().numpy().astype("uint32")
        for r in range(world):
            f_r, c_r = C.c_int(), C.c_int()
            L.achip_shard_bounds(n, world, r, C.byref(f_r), C.byref(c_r))
            for i in range(c_r.value):
                s = r * slots + i
                assert (len3 is None) == (took == 1) and (len3 is None or len3[s] == int(l3[s]))
                assert wh[off3[s]:off3[s] + int(l3[s])].tobytes() == exp[f_r.value + i], f"rank {rank}: form {took} frame {f_r.value + i} differs"
    os.environ.pop("ASCIICHAT_HIP_GATHER", None)
    plan.close()
    # ---- (2b) frames of configs[4]'s size through the packed gather: 1.8 MB each, one per rank plus one (uneven shards), so
    # that a rank's block spans several of the transport's chunks and the lengths-first sizing sees megabytes
    if full:
        nb, Wb, Hb = world + 1, 400, 120
        big = [orc.frame_hash_noise(400, 240, 900 + i) for i in range(nb)]
        slots_b = L.achip_shard_slots(nb, world)
        fb, cb = C.c_int(), C.c_int()
        L.achip_shard_bounds(nb, world, rank, C.byref(fb), C.byref(cb))
        devb = [torch.from_numpy(big[fb.value + i]).cuda() for i in range(cb.value)]
        fr = [pkg.frame_setup(d.data_ptr(), 400, 240, Wb, Hb, 2, False, False, False) for d in devb]
        assert fr, "world + 1 frames: every rank owns at least one"
        planb = pkg.Plan(pkg.MODE_HB_TRUE, orc.PALETTE_STANDARD, fr)
        sb = planb.stride
        slabb = torch.zeros(world * slots_b * sb, dtype=torch.uint8, device="cuda")
        lnb = torch.zeros(world * slots_b, dtype=torch.int32, device="cuda")
        planb.render(slabb.data_ptr() + rank * slots_b * sb, sb, lnb.data_ptr() + 4 * rank * slots_b, st)
        packedb = torch.zeros(world * slots_b * sb, dtype=torch.uint8, device="cuda")
        offb, plenb, blkb = comm.all_gather_packed(slabb.data_ptr(), sb, lnb.data_ptr(), slots_b, packedb.data_ptr(), slots_b * sb, st)
        torch.cuda.synchronize()
        pb_h = packedb.cpu().numpy()
        assert blkb % 16 == 0 and blkb <= slots_b * sb
        for r in range(world):
            f_r, c_r = C.c_int(), C.c_int()
            L.achip_shard_bounds(nb, world, r, C.byref(f_r), C.byref(c_r))
            for i in range(c_r.value):
                sl = r * slots_b + i
                e = orc.convert_with_caps(big[f_r.value + i], Wb, Hb, 3, 2, False, False, False)
                assert plenb[sl] == len(e) > 1_000_000, (plenb[sl], len(e))
                assert pb_h[offb[sl]:offb[sl] + plenb[sl]].tobytes() == e, f"rank {rank}: big packed frame {f_r.value + i} differs"
        planb.close()
    # ---- (3) the pixel-space grid from tiles that live on different ranks -----------------------------------------------
    for n_src, has_video in ((9, None), (5, [True, True, False, True, True]), (3, None)):
        srcs = [orc.frame_hash_noise(320, 180, 40 + k) if k % 2 else orc.frame_bars(320, 180, k) for k in range(n_src)]
        grid = pkg.Grid(comm, [(320, 180)] * n_src, 80, 24, has_video)
        own = [k for k in range(n_src) if grid.owner(k) == rank]
        assert sorted(own) == list(range(own[0], own[0] + len(own))) if own else True
        dsrc = {k: torch.from_numpy(srcs[k]).cuda() for k in own}  # a rank only holds the sources it owns
        grid.exchange({k: t.data_ptr() for k, t in dsrc.items()}, st)
        live = [s if (has_video is None or has_video[k]) else None for k, s in enumerate(srcs)]
        ref = orc.composite(live, 80, 24)
        for mode, (cl, rm) in ((pkg.MODE_TRUE_FG, (3, 0)), (pkg.MODE_HB_TRUE, (3, 2))):
            h = 48 if rm == 2 else 24
            f = pkg.frame_setup(None, 80, 48, 80, h, rm, True, True, False)
            f.comp = grid.composite_dev
            plan = pkg.Plan(mode, orc.PALETTE_STANDARD, [f, f])
            out = torch.zeros(2 * plan.stride, dtype=torch.uint8, device="cuda")
            l2 = torch.zeros(2, dtype=torch.int32, device="cuda")
            plan.render(out.data_ptr(), plan.stride, l2.data_ptr(), st)
            torch.cuda.synchronize()
            e = orc.convert_with_caps(ref, 80, h, cl, rm, True, True, False)
            for i in range(2):
                got = out[i * plan.stride:i * plan.stride + int(l2[i].item())].cpu().numpy().tobytes()
                assert got == e, f"rank {rank}: grid of {n_src} sources, mode {mode}, target {i} differs"
            plan.close()
        grid.close()
    comm.close()
    print(f"ok rank {rank}")


if __name__ == "__main__":
    main()
