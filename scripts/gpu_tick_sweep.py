#!/usr/bin/env python3
"""The end-to-end server tick of bench.py (tick_e2e) on its own: sampled-image ingest with the environment's ingest
settings (ASCIICHAT_HIP_INGEST_THREADS / _ZERO_COPY / _SPIN_US).  Prints one JSON object.  GPU box only."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import bench  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
torch.cuda.set_device(0)
forms = tuple(sys.argv[1].split(",")) if len(sys.argv) > 1 else ("sampled_pixels_batched", "sampled_images")
out = bench.tick_e2e(torch, pkg, forms=forms)
out["env"] = {k: v for k, v in os.environ.items() if k.startswith("ASCIICHAT_HIP_")}
print(json.dumps(out))
