#!/usr/bin/env python3
"""sha256[:16] over the sources that decide what the kernels are (csrc/, include/, the Makefile): the identity a committed
profile is tagged with, so that bench.py can tell a profile of THIS code from a stale one (frac_profile)."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_id():
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "ascii-chat_amd", "csrc", "*")) + glob.glob(os.path.join(ROOT, "include", "*.h")) +
                   [os.path.join(ROOT, "ascii-chat_amd", "Makefile")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_id())
