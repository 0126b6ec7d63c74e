import os, sys, statistics
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import bench, orc
from __graft_entry__ import load_package
pkg = load_package(); torch.cuda.set_device(0); cur = torch.cuda.current_stream()
def time_plan(plan, n, reps=300):
    out = torch.empty(n * plan.stride, dtype=torch.uint8, device="cuda"); ln = torch.zeros(n, dtype=torch.int32, device="cuda")
    for _ in range(20): plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), cur.cuda_stream)
    torch.cuda.synchronize(); ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        for _ in range(reps): plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), cur.cuda_stream)
        e1.record(cur); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return statistics.median(ts), out, ln.cpu().numpy().astype("uint32")
CASES = [(3840, 2160, 200, 60, 1), (3840, 2160, 200, 60, 2), (3840, 2160, 200, 60, 8), (1920, 1080, 120, 40, 1), (1920, 1080, 120, 40, 16), (1920, 1080, 80, 24, 32)]
if len(sys.argv) > 1 and sys.argv[1] == 'small':
    CASES = [(640, 480, 80, 24, 1), (640, 480, 80, 24, 8), (1920, 1080, 160, 48, 1), (1920, 1080, 160, 48, 9), (3840, 2160, 200, 60, 1), (3840, 2160, 400, 120, 1), (3840, 2160, 400, 120, 4)]
if len(sys.argv) > 1 and sys.argv[1] == 'mid':
    CASES = [(3840, 2160, 200, 60, 16), (3840, 2160, 200, 60, 32), (3840, 2160, 200, 60, 64), (1920, 1080, 120, 40, 48), (1920, 1080, 120, 40, 100), (1920, 1080, 160, 48, 32), (1920, 1080, 80, 24, 100)]
for (sw, sh, W, H, nb) in CASES:
    imgs = bench.make_frames(torch, nb, sw, sh, 5)
    fr = [pkg.frame_setup(imgs.data_ptr() + i * sw * sh * 3, sw, sh, W, H, 0, False, False, False) for i in range(nb)]
    exp = orc.convert_with_caps(np.ascontiguousarray(imgs[0].cpu().numpy()), W, H, 3, 0, False, False, False)
    host0 = np.ascontiguousarray(imgs[0].cpu().numpy())
    RUN = len(sys.argv) > 1 and sys.argv[-1] == "run"  # the run-structured modes instead (rows kernel)
    for mode, nm, cl, rm in (((0, "mono", 0, 0), (5, "half-block truecolor", 3, 2), (6, "half-block 256", 2, 2)) if RUN else ((1, "truecolor", 3, 0), (2, "ansi256", 2, 0))):
        fr_m = fr if rm == 0 else [pkg.frame_setup(imgs.data_ptr() + i * sw * sh * 3, sw, sh, W, H, rm, False, False, False) for i in range(nb)]
        want = orc.convert_with_caps(host0, W, H, cl, rm, False, False, False)
        plan = pkg.Plan(mode, bench.PALETTE_STANDARD, fr_m)
        t, out, lens = time_plan(plan, nb)
        got = bytes(out[:int(lens[0])].cpu().numpy())
        print(f"{nb:3d} x ({sw}x{sh} -> {W}x{H} {nm}): {t:7.2f} us  variant {plan.variant} parts {plan.parts} ok {got == want}")
        plan.close()
