#!/usr/bin/env python3
"""Randomised soak on the GPU (not part of the pytest suites): many random batches -- sizes, modes, palettes, paddings,
display ops, batch sizes, band splits, kernel geometries -- every frame compared byte-for-byte with the oracle.
Usage (GPU box): python scripts/gpu_soak.py [--rounds 150] [--seed 1]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402
from achip_ctypes import ALL_MODES, MODE_CAPS, MODE_NAMES, MODE_TRUE_BG  # noqa: E402
from test_random_differential import PALETTES, oracle_case, random_image  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=150)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import torch

    from __graft_entry__ import load_package

    pkg = load_package()
    torch.cuda.set_device(0)
    rng = np.random.default_rng(args.seed)
    checked = 0
    for rnd in range(args.rounds):
        mode = int(rng.choice(ALL_MODES))
        palette = PALETTES[int(rng.integers(0, len(PALETTES)))]
        aspect = bool(rng.integers(0, 2)) and mode != MODE_TRUE_BG
        pad = bool(rng.integers(0, 2))
        rm = MODE_CAPS.get(mode, (3, 0))[1]
        nframes = int(rng.choice([1, 2, 3, 7, 16, 40, 100, 200, 300, 420]))
        big = rng.integers(0, 4) == 0
        wide = rng.integers(0, 3) == 0
        flt = int(rng.choice([0, 0, 0, 3, 7, 11]))
        fx, fy = bool(rng.integers(0, 4) == 0), bool(rng.integers(0, 4) == 0)
        pool = []
        for _ in range(min(nframes, 12)):  # a few distinct sources, reused round-robin
            sw, sh = (int(rng.integers(200, 2000)), int(rng.integers(100, 1100))) if big else (int(rng.integers(1, 300)), int(rng.integers(1, 200)))
            img = random_image(rng, sw, sh)
            pool.append((img, torch.from_numpy(np.ascontiguousarray(img)).cuda()))
        frames, cases = [], []
        for k in range(nframes):
            img, dev = pool[k % len(pool)]
            W, H = (int(rng.integers(1, 520)), int(rng.integers(1, 140))) if big else (int(rng.integers(1, 200)), int(rng.integers(1, 70)))
            if big and wide:  # round 6: rows beyond one block of the rows kernel (cut into segments, geometries 27 / 29)
                W, H = int(rng.integers(449, 1500)), int(rng.integers(1, 40))
            f = pkg.frame_setup(dev.data_ptr(), img.shape[1], img.shape[0], W, H, rm, pad, aspect, False)
            if f is None:
                continue
            if (flt or fx or fy) and mode != MODE_TRUE_BG:
                import ctypes as C
                assert pkg.lib().achip_frame_set_display_ops(C.byref(f), fx, fy, flt) == 0
            frames.append(f)
            cases.append((img, W, H))
        if not frames:
            continue
        plan = pkg.Plan(mode, palette, frames)
        choice = int(rng.integers(0, 8))
        try:
            if choice == 1:
                plan.set_split(-1)
            elif choice == 2:
                plan.set_split(int(rng.integers(1, 9)))
            elif choice == 3:
                plan.set_variant(int(rng.choice([0, 1, 2, 4])))
            elif choice == 4:
                plan.set_variant(int(rng.choice([1, 2, 4])))
                plan.set_split(int(rng.integers(1, 6)))
            elif choice >= 5:  # the wave-autonomous kernels: stream geometries (per-cell modes) / rows geometries (run-structured
                # modes); refused for the other family and for rows wider than a block, with the fused CRC forced on
                plan.set_variant(int(rng.choice([24, 25, 25, 26, 27, 29]) if mode in (0, 5, 6, 7, 8) else rng.choice([16, 17, 17, 18, 19])))
                plan.set_fused_crc(1)
        except RuntimeError:
            pass  # geometry cannot hold this batch's rows: keep the automatic one
        out = torch.full((len(frames) * plan.stride,), 0xAB, dtype=torch.uint8, device="cuda")
        ln = torch.zeros(len(frames), dtype=torch.int32, device="cuda")
        for _ in range(2):
            plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), torch.cuda.current_stream().cuda_stream)
        # wire stage over the same slab: CRC-32C, packet headers, packet CRCs
        nfr = len(frames)
        dims_t = torch.tensor([[c[1], c[2]] for c in cases], dtype=torch.int32, device="cuda")
        crc_t = torch.zeros(nfr, dtype=torch.int32, device="cuda")
        hdr_t = torch.zeros(nfr * 24, dtype=torch.uint8, device="cuda")
        pkt_t = torch.zeros(nfr, dtype=torch.int32, device="cuda")
        assert pkg.lib().asciichat_hip_frame_packets(out.data_ptr(), plan.stride, ln.data_ptr(), plan.stride, nfr, dims_t.data_ptr(),
                                                     crc_t.data_ptr(), hdr_t.data_ptr(), pkt_t.data_ptr(),
                                                     torch.cuda.current_stream().cuda_stream) == 0, pkg.last_error()
        packed_t = torch.full((len(frames) * plan.stride,), 0xCD, dtype=torch.uint8, device="cuda")
        off_t = torch.zeros(nfr + 1, dtype=torch.int64, device="cuda")
        pkg.pack_frames(out.data_ptr(), plan.stride, ln.data_ptr(), nfr, packed_t.data_ptr(), packed_t.numel(), off_t.data_ptr(), None,
                        torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        host, lens = out.cpu().numpy(), ln.cpu().numpy().astype(np.uint32)
        ph, oh = packed_t.cpu().numpy(), off_t.cpu().numpy()
        for k in range(nfr):
            assert oh[k] % 16 == 0 and (ph[oh[k]:oh[k] + int(lens[k])] == host[k * plan.stride:k * plan.stride + int(lens[k])]).all(), ("pack", rnd, k)
        assert int(oh[nfr]) == sum((int(v) + 15) // 16 * 16 for v in lens), ("pack total", rnd)
        crc_h, pkt_h, hdr_h = crc_t.cpu().numpy().astype(np.uint32), pkt_t.cpu().numpy().astype(np.uint32), hdr_t.cpu().numpy()
        memo = {}
        for k, (img, W, H) in enumerate(cases):
            assert lens[k] < 0xFFFFFFF0, (rnd, k, hex(int(lens[k])))
            got = host[k * plan.stride:k * plan.stride + int(lens[k])].tobytes()
            key = (id(img), W, H)
            if key not in memo:
                if (flt or fx or fy) and mode != MODE_TRUE_BG:
                    cl, rmm = MODE_CAPS[mode]
                    memo[key] = orc.display_convert(img, W, H, cl, rmm, pad, aspect, fx, fy, flt, palette)
                else:
                    memo[key] = oracle_case(img, W, H, mode, aspect, pad, palette)
            if got != memo[key]:  # keep the case for a replay under the emulator (tests/emu.py) before failing
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                np.savez(os.path.join(ROOT, "gpurun_out", "soak_failure.npz"), mode=mode, palette=palette, pad=pad, aspect=aspect, k=k,
                         variant=plan.variant, parts=plan.parts, fused_crc=int(plan.fused_crc), choice=choice, flt=flt, fx=fx, fy=fy,
                         dims=np.array([[c[1], c[2]] for c in cases]), pool_index=np.array([j % len(pool) for j in range(len(cases))]),
                         got=np.frombuffer(got, np.uint8), exp=np.frombuffer(memo[key], np.uint8),
                         **{f"img{j}": p_[0] for j, p_ in enumerate(pool)})
            assert got == memo[key], (rnd, MODE_NAMES[mode], k, img.shape, W, H, pad, aspect, palette, plan.variant, plan.parts, flt, fx, fy)
            if k < 24:  # the bit-serial oracle CRC is slow: a sample per batch
                eh, ep = orc.ascii_frame_packet(got, W, H)
                assert int(crc_h[k]) == orc.crc32c(got) and hdr_h[24 * k:24 * k + 24].tobytes() == eh and int(pkt_h[k]) == ep, ("crc", rnd, k, len(got))
            checked += 1
        # render + frame CRC in one go (fused into the stream kernel where the plan's geometry carries it, the
        # stand-alone kernel behind the render otherwise): same bytes, same checksums, same headers
        out_c = torch.full((len(frames) * plan.stride,), 0x5C, dtype=torch.uint8, device="cuda")
        ln_c = torch.zeros(nfr, dtype=torch.int32, device="cuda")
        crc_c = torch.full((nfr,), 0x7E7E7E7E, dtype=torch.int32, device="cuda")
        hdr_c = torch.zeros(nfr * 24, dtype=torch.uint8, device="cuda")
        pkt_c = torch.zeros(nfr, dtype=torch.int32, device="cuda")
        st_c = torch.cuda.current_stream().cuda_stream
        if rnd % 2:  # everything in the render launch (where the geometry carries the fused CRC)
            plan.render_packets(out_c.data_ptr(), plan.stride, ln_c.data_ptr(), dims_t.data_ptr(), crc_c.data_ptr(),
                                hdr_c.data_ptr(), pkt_c.data_ptr(), st_c)
        else:
            plan.render_crc(out_c.data_ptr(), plan.stride, ln_c.data_ptr(), crc_c.data_ptr(), st_c)
            assert pkg.lib().asciichat_hip_packets_from_crc(ln_c.data_ptr(), crc_c.data_ptr(), nfr, dims_t.data_ptr(), hdr_c.data_ptr(),
                                                            pkt_c.data_ptr(), st_c) == 0, pkg.last_error()
        torch.cuda.synchronize()
        lens_c = ln_c.cpu().numpy().astype(np.uint32)
        assert (lens_c == lens).all(), ("render_crc lengths", rnd, plan.variant)
        host_c = out_c.cpu().numpy()
        for k in range(nfr):
            assert host_c[k * plan.stride:k * plan.stride + int(lens[k])].tobytes() == host[k * plan.stride:k * plan.stride + int(lens[k])].tobytes(), ("render_crc bytes", rnd, k)
        assert (crc_c.cpu().numpy().astype(np.uint32) == crc_h).all(), ("fused crc", rnd, plan.variant, plan.fused_crc)
        assert (pkt_c.cpu().numpy().astype(np.uint32) == pkt_h).all() and (hdr_c.cpu().numpy() == hdr_h).all(), ("fused packets", rnd)
        fused_rounds = locals().get("fused_rounds", 0) + int(plan.fused_crc)
        # frames at their exact lengths straight from the render kernel (per-cell plans whose frames fit the LDS image):
        # every frame's bytes at its offset, the destination tiled in some order, checksums / headers / packet CRCs
        if plan.exact_length:
            exact_rounds = locals().get("exact_rounds", 0) + 1
            plan.set_exact_length(1)
            cap_x = nfr * plan.stride
            dst_x = torch.full((cap_x,), 0x3C, dtype=torch.uint8, device="cuda")
            off_x = torch.zeros(nfr + 1, dtype=torch.int64, device="cuda")
            pl_x = torch.zeros(nfr, dtype=torch.int32, device="cuda")
            ln_x = torch.zeros(nfr, dtype=torch.int32, device="cuda")
            crc_x = torch.full((nfr,), 0x7E7E7E7E, dtype=torch.int32, device="cuda")
            hdr_x = torch.zeros(nfr * 24, dtype=torch.uint8, device="cuda")
            pkt_x = torch.zeros(nfr, dtype=torch.int32, device="cuda")
            for _ in range(2):
                if rnd % 2:
                    plan.render_packets_packed(0, plan.stride, ln_x.data_ptr(), dims_t.data_ptr(), crc_x.data_ptr(), hdr_x.data_ptr(),
                                               pkt_x.data_ptr(), dst_x.data_ptr(), cap_x, off_x.data_ptr(), pl_x.data_ptr(), st_c)
                else:
                    plan.render_packed(0, plan.stride, ln_x.data_ptr(), dst_x.data_ptr(), cap_x, off_x.data_ptr(), pl_x.data_ptr(), st_c)
            torch.cuda.synchronize()
            dx, ox, px = dst_x.cpu().numpy(), off_x.cpu().numpy(), pl_x.cpu().numpy().astype(np.uint32)
            assert (px == lens).all() and (ln_x.cpu().numpy().astype(np.uint32) == lens).all(), ("exact lengths", rnd)
            spans = sorted((int(ox[k]), int(ox[k]) + (int(lens[k]) + 15) // 16 * 16) for k in range(nfr))
            assert spans[0][0] == 0 and all(spans[k][1] == spans[k + 1][0] for k in range(nfr - 1)) and spans[-1][1] == int(ox[nfr]), ("exact tiling", rnd)
            for k in range(nfr):
                assert dx[int(ox[k]):int(ox[k]) + int(lens[k])].tobytes() == host[k * plan.stride:k * plan.stride + int(lens[k])].tobytes(), ("exact bytes", rnd, k)
            if rnd % 2:
                assert (crc_x.cpu().numpy().astype(np.uint32) == crc_h).all(), ("exact crc", rnd, plan.variant)
                assert (pkt_x.cpu().numpy().astype(np.uint32) == pkt_h).all() and (hdr_x.cpu().numpy() == hdr_h).all(), ("exact packets", rnd)
            plan.set_exact_length(-1)
        # the same plan after an update to new terminal sizes (the tick when clients resize), rendered as two
        # sub-ranges into the slab (what the ranks of a sharded batch do)
        if rnd % 3 == 0 and (flt or fx or fy) == 0:
            frames2, cases2 = [], []
            for (img, _, _), f_old in zip(cases, frames):
                W2, H2 = (int(rng.integers(1, 200)), int(rng.integers(1, 70)))
                f2 = pkg.frame_setup(f_old.src, img.shape[1], img.shape[0], W2, H2, rm, pad, aspect, False)
                if f2 is None:
                    f2, W2, H2 = f_old, None, None
                frames2.append(f2)
                cases2.append((img, W2, H2))
            try:
                plan.update(frames2, torch.cuda.current_stream().cuda_stream)
            except RuntimeError:
                frames2 = None  # the forced geometry cannot hold the new rows: a caller would rebuild the plan
            if frames2 is not None:
                out2 = torch.full((len(frames2) * plan.stride,), 0xCD, dtype=torch.uint8, device="cuda")
                half = len(frames2) // 2
                st2 = torch.cuda.current_stream().cuda_stream
                plan.render(out2.data_ptr(), plan.stride, ln.data_ptr(), st2, 0, half)
                plan.render(out2.data_ptr() + half * plan.stride, plan.stride, ln.data_ptr() + 4 * half, st2, half, len(frames2) - half)
                torch.cuda.synchronize()
                host2, lens2 = out2.cpu().numpy(), ln.cpu().numpy().astype(np.uint32)
                for k, (img, W2, H2) in enumerate(cases2):
                    if W2 is None:
                        continue
                    assert lens2[k] < 0xFFFFFFF0
                    got2 = host2[k * plan.stride:k * plan.stride + int(lens2[k])].tobytes()
                    assert got2 == oracle_case(img, W2, H2, mode, aspect, pad, palette), ("update", rnd, k, W2, H2, plan.variant, plan.parts)
                    checked += 1
        print(f"round {rnd:3d}: {MODE_NAMES[mode]:10s} frames {len(frames):3d} geometry v{plan.variant} bands {plan.parts:3d} "
              f"{'crc fused' if plan.fused_crc else 'crc separate'} ok", flush=True)
        plan.close()
    # pixel-space composites (the server's multi-source grid): random source counts / sizes / terminal sizes,
    # rendered fused (canvas never built) for several client modes at once
    import ctypes as C
    comp_checked = 0
    for rnd in range(max(1, args.rounds // 4)):
        n_src = int(rng.integers(1, 10))
        tw, th = int(rng.integers(20, 260)), int(rng.integers(8, 90))
        imgs = [random_image(rng, int(rng.integers(16, 900)), int(rng.integers(16, 500))) for _ in range(n_src)]
        dev = [torch.from_numpy(np.ascontiguousarray(i)).cuda() for i in imgs]
        ptrs = (C.c_void_p * n_src)(*[d.data_ptr() for d in dev])
        ws = (C.c_int * n_src)(*[i.shape[1] for i in imgs])
        hs = (C.c_int * n_src)(*[i.shape[0] for i in imgs])
        comp = pkg.Composite()
        pkg.lib().achip_composite_setup(C.byref(comp), ptrs, ws, hs, n_src, tw, th)
        comp_dev = C.c_void_p()
        assert pkg.lib().asciichat_hip_composite_upload(C.byref(comp), C.byref(comp_dev)) == 0
        canvas = orc.composite(imgs, tw, th)
        mode = int(rng.choice([0, 1, 2, 3, 5, 6, 8]))
        cl, rm = MODE_CAPS[mode]
        pad = bool(rng.integers(0, 2))
        h = 2 * th if rm == 2 else th
        nclients = int(rng.choice([1, 3, 9, 40]))
        frames = []
        for _ in range(nclients):
            f = pkg.frame_setup(None, tw, 2 * th, tw, h, rm, pad, True, False)
            assert f is not None
            f.comp = comp_dev.value
            frames.append(f)
        plan = pkg.Plan(mode, orc.PALETTE_STANDARD, frames)
        if rng.integers(0, 3) == 0:
            plan.set_split(-1)
        out = torch.zeros(nclients * plan.stride, dtype=torch.uint8, device="cuda")
        ln = torch.zeros(nclients, dtype=torch.int32, device="cuda")
        plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        exp = orc.convert_with_caps(canvas, tw, h, cl, rm, pad, True, False)
        host, lens = out.cpu().numpy(), ln.cpu().numpy().astype(np.uint32)
        for k in range(nclients):
            got = host[k * plan.stride:k * plan.stride + int(lens[k])].tobytes()
            assert got == exp, ("composite", rnd, MODE_NAMES[mode], n_src, tw, th, pad, k, plan.variant, plan.parts)
            comp_checked += 1
        print(f"composite {rnd:3d}: {MODE_NAMES[mode]:10s} sources {n_src} term {tw}x{th} clients {nclients:2d} v{plan.variant} bands {plan.parts:3d} ok", flush=True)
        plan.close()
        pkg.lib().asciichat_hip_free(comp_dev)
    print(f"soak OK: {checked} frames + {comp_checked} composite frames byte-identical to the oracle; "
          f"{locals().get('fused_rounds', 0)} batches with the checksum fused into the render, {locals().get('exact_rounds', 0)} "
          "batches also written at their exact lengths by the render kernel (tiling, bytes, checksums, headers checked)")


if __name__ == "__main__":
    main()
