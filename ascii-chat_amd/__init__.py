"""ascii-chat_amd: Python-side plumbing for libasciichat_hip.so (the MI355X-native render path).

The product is the C-ABI shared library built from csrc/ (see include/asciichat_hip.h and
include/asciichat_render.h); this package only loads it through ctypes for the test-suite and bench.py
and uses torch for device memory, streams and torch.distributed.  There is no Python or CPU
implementation of the render path here: if the library or a GPU is missing, calls fail loudly.

The directory name contains a hyphen, so import it with the helper in __graft_entry__.py:
    from __graft_entry__ import load_package; achip = load_package()
"""
from .binding import *  # noqa: F401,F403
from . import distributed  # noqa: F401
