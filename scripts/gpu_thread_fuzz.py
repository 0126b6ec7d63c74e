#!/usr/bin/env python3
"""Concurrency soak on the GPU box: N threads call the drop-in entry points at once (ctypes releases the GIL), with
more distinct palettes than the glyph-table cache holds, so that per-thread contexts, the palette caches (host and
device, with LRU recycling of unpinned entries) and the pinned pool are exercised under contention."""
import ctypes as C
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402


def main():
    import torch  # noqa: F401

    from __graft_entry__ import load_package

    pkg = load_package()
    L = pkg.lib()
    nthreads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    alphabet = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789"
    imgs = [orc.frame_hash_noise(64 + 16 * k, 48 + 8 * k, k) for k in range(6)]
    orc.lib()
    errors = []

    def worker(tid):
        try:
            rng = np.random.default_rng(100 + tid)
            caps = pkg.TermCaps()
            caps.utf8_support = True
            for it in range(calls):
                k = int(rng.integers(0, 3000))
                pal = ("  " + alphabet[k % 62] + alphabet[(k // 62) % 62] + alphabet[(k * 7) % 62] + "#").encode()
                img = imgs[int(rng.integers(0, len(imgs)))]
                arr = np.ascontiguousarray(img)
                im = pkg.Image(arr.shape[1], arr.shape[0], arr.ctypes.data, 0)
                cl, rm = int(rng.choice([0, 1, 2, 3])), int(rng.choice([0, 2]))
                caps.color_level, caps.render_mode = cl, rm
                W, H = int(rng.integers(1, 60)), int(rng.integers(1, 20))
                got = pkg.take_string(L.ascii_convert_with_capabilities(C.byref(im), W, H, C.byref(caps), False, False, pal))
                if it % 10 == 0:
                    exp = orc.convert_with_caps(img, W, H, cl, rm, False, False, False, pal)
                    assert got == exp, (tid, it, W, H, cl, rm, pal)
                else:
                    assert got is not None and len(got) > 0
                if it % 97 == 0:  # pool traffic from several threads
                    p = L.buffer_pool_alloc(None, 5 << 20)
                    assert p
                    L.buffer_pool_free(None, p, 5 << 20)
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))

    L.buffer_pool_alloc.restype = C.c_void_p
    L.buffer_pool_alloc.argtypes = [C.c_void_p, C.c_size_t]
    L.buffer_pool_free.restype = None
    L.buffer_pool_free.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    ts = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:3]
    print(f"thread fuzz OK: {nthreads} threads x {calls} drop-in calls, 3000 palettes, every 10th render compared with the oracle")


if __name__ == "__main__":
    main()
