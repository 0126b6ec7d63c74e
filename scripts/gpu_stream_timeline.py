"""Per-wave timeline of the stream kernel (render_stream.hpp, diagnostics stamps): one 256-frame launch at a time,
fresh input batch per launch.  Prints, per stamp, the distribution over all waves of (stamp - earliest kernel entry)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

NAMES = ["entry", "barrier passed (after request)", "samples requested", "samples arrived", "tokens+scan", "look-back",
         "token bytes in LDS", "stores issued"]


def main():
    pkg = load_package()
    torch.cuda.set_device(0)
    name = sys.argv[1] if len(sys.argv) > 1 else "1080p_80x24_truecolor"
    variant = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    with_crc = len(sys.argv) > 3 and sys.argv[3] == "crc"  # the instantiation that carries the frame CRC
    if with_crc:
        NAMES[6], NAMES[7] = "stores issued", "block checksummed + placed"
    sw, sh, W, H, cl, rm = bench.WORKLOADS[name]
    B = int(os.environ.get("TIMELINE_BATCH", "256"))  # 1: what a lone frame's workgroup does on an idle GPU
    sets = [bench.make_frames(torch, B, sw, sh, 900 + s) for s in range(6)]
    plans = [bench.build_plan(pkg, t, W, H, cl, rm)[0] for t in sets]
    for p in plans:
        if variant >= 0:
            p.set_variant(variant)
    variant = plans[0].variant  # -1 on the command line: the automatic choice (small launches: PARTS of geometry 18)
    parts = max(1, plans[0].parts) if variant >= 16 else 1
    waves = pkg.lib().achip_variant_block(variant) // 64
    B *= parts  # one row of stamps per WORKGROUP
    out = torch.empty((B // parts) * plans[0].stride, dtype=torch.uint8, device="cuda")
    ln = torch.zeros(B // parts, dtype=torch.int32, device="cuda")
    prof = torch.zeros(B * waves * 8, dtype=torch.int64, device="cuda")
    crc = torch.zeros(B, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    acc = []
    for k in range(12):
        prof.zero_()
        torch.cuda.synchronize()
        if with_crc:
            rc = pkg.lib().asciichat_hip_plan_render_crc_profiled(plans[k % len(plans)]._h, out.data_ptr(), plans[0].stride,
                                                                  ln.data_ptr(), crc.data_ptr(), prof.data_ptr(), st)
            assert rc == 0, pkg.last_error()
        else:
            plans[k % len(plans)].render_profiled(out.data_ptr(), plans[0].stride, ln.data_ptr(), prof.data_ptr(), st)
        torch.cuda.synchronize()
        if k >= 4:
            acc.append(prof.cpu().numpy().reshape(B, waves, 8).astype(np.int64))
    if B == parts:  # every wave of the last launch, stamp by stamp
        a = acc[-1]
        t0 = a[:, :, 0][a[:, :, 0] != 0].min()
        print(f"# {name} variant {variant} parts {parts}, ONE frame: per workgroup.wave, us since the first wave's entry")
        print("# wg.wave " + " ".join(f"{n[:12]:>12s}" for n in NAMES))
        for g in range(a.shape[0]):
            for w in range(waves):
                if a[g, w, 7]:
                    print(f"  {g:3d}.{w:<3d} " + " ".join(f"{(a[g, w, s] - t0) / 100.0:12.2f}" for s in range(8)))
    print(f"# {name} variant {variant}{' + fused frame CRC' if with_crc else ''}: {B} frame(s), {waves} waves per frame, 100 MHz wall clock -> us; per stamp over all active waves of 8 launches")
    print(f"# {'stamp':28s} {'min':>7s} {'p10':>7s} {'median':>7s} {'p90':>7s} {'max':>7s}")
    rows = [[] for _ in range(8)]
    for a in acc:
        active = a[:, :, 7] != 0
        t0 = a[:, :, 0][a[:, :, 0] != 0].min()
        for s in range(8):
            v = a[:, :, s][active]
            rows[s].append((v - t0) / 100.0)
    for s in range(8):
        v = np.concatenate(rows[s])
        print(f"  {NAMES[s]:28s} {v.min():7.2f} {np.percentile(v, 10):7.2f} {np.median(v):7.2f} {np.percentile(v, 90):7.2f} {v.max():7.2f}")


if __name__ == "__main__":
    main()
