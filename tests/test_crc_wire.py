"""Wire stage after render: CRC-32C + ascii_frame_packet_t header, kernels under the emulator.
The oracle's bitwise CRC is pinned on the reference's own known answers (tests/unit/network/crc32_hw_test.c)."""
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import emu  # noqa: E402
import orc  # noqa: E402


def test_oracle_crc_known_answers():
    assert orc.crc32c(b"") == 0                               # crc32_hw_test.c:14-21
    assert orc.crc32c(b"Hello, World!") == 0x4D551068         # crc32_hw_test.c:32-50
    assert orc.crc32c(b"123456789") == 0xE3069283             # the CRC-32C check value (RFC 3720 B.4 family)
    assert orc.crc32c(bytes(32)) == 0x8A9136AA                # RFC 3720 B.4: 32 bytes of zeros
    assert orc.crc32c(b"\xff" * 32) == 0x62A8AB43             # RFC 3720 B.4: 32 bytes of ones
    assert orc.crc32c(bytes(range(32))) == 0x46DD794E         # RFC 3720 B.4: incrementing
    assert orc.crc32c(b"a") != orc.crc32c(b"b") and orc.crc32c(b"ab") != orc.crc32c(b"ba")  # :245-273
    hdr, pkt = orc.ascii_frame_packet(b"Hello, World!", 80, 24)
    assert hdr == struct.pack(">6I", 80, 24, 13, 0, 0x4D551068, 0)
    assert pkt == orc.crc32c(hdr + b"Hello, World!")


def test_gf2_constants():
    L = emu.lib()
    X0, X8, XINV8 = 0x80000000, 0x00800000, 0xFDE39562
    assert L.emu_crc_mulmod(X8, XINV8) == X0
    for k in range(32):
        assert L.emu_crc_x8_pow2(k) == L.emu_crc_pow(X8, 1 << k), k
    # shifting the register by n zero bytes is a multiplication by x^(8n)
    msg = b"The quick brown fox jumps over the lazy dog"
    raw = orc.crc32c(msg) ^ 0xFFFFFFFF ^ L.emu_crc_mulmod(0xFFFFFFFF, L.emu_crc_pow(X8, len(msg)))
    raw_pad = orc.crc32c(msg + bytes(37)) ^ 0xFFFFFFFF ^ L.emu_crc_mulmod(0xFFFFFFFF, L.emu_crc_pow(X8, len(msg) + 37))
    assert L.emu_crc_mulmod(raw, L.emu_crc_pow(X8, 37)) == raw_pad


def _frames():
    rng = np.random.default_rng(5)
    img = orc.frame_torture()
    frames = [orc.convert_with_caps(img, 80, 24, 3, 0), orc.convert_with_caps(img, 80, 24, 2, 0),
              orc.convert_with_caps(img, 33, 7, 0, 0), b"", b"x", b"Hello, World!", bytes(15), bytes(16), bytes(17),
              rng.integers(0, 256, 4095, dtype=np.uint8).tobytes(), rng.integers(0, 256, 4096, dtype=np.uint8).tobytes(),
              rng.integers(0, 256, 4097, dtype=np.uint8).tobytes(), rng.integers(0, 256, 70001, dtype=np.uint8).tobytes()]
    dims = [(80, 24), (80, 24), (33, 7)] + [(i + 1, 2 * i + 1) for i in range(len(frames) - 3)]
    return frames, dims


@pytest.fixture(params=[0, 1], ids=["spans+finish", "one launch"])
def one_launch(request):
    """the span form with crc32c_finish_kernel behind it (the stand-alone entry points) / finishing its frames itself: the
    last span of a frame to arrive combines the registers (what a plan's wire pass launches; round 4, session 2)"""
    emu.lib().emu_set_crc_one_launch(request.param)
    yield request.param
    emu.lib().emu_set_crc_one_launch(0)


@pytest.mark.parametrize("force", [None, (3, 6), (2, 16), (70, 1), (130, 1)],
                         ids=["launcher", "3x6", "2x16", "70x1", "130x1"])
def test_crc_and_packet_headers_emulated(force, one_launch):
    frames, dims = _frames()
    crc, hdr, pkt = emu.crc32c_frames(frames, dims, force=force)
    for i, f in enumerate(frames):
        assert int(crc[i]) == orc.crc32c(f), (i, len(f))
        eh, ep = orc.ascii_frame_packet(f, *dims[i])
        assert hdr[i] == eh, i
        assert int(pkt[i]) == ep, i
    crc2, _, _ = emu.crc32c_frames(frames, None, force=force, want_headers=False)
    assert (crc2 == crc).all()


def test_large_buffer_spans_emulated(one_launch):
    # > 128 KB: the launcher cuts the buffer into 64 KB spans finished by the second kernel (ingest payload sizes)
    rng = np.random.default_rng(9)
    for n in (131073, 300000, 640 * 480 * 3):
        buf = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        crc, _, _ = emu.crc32c_frames([buf, buf[:n // 2]], None, want_headers=False)
        assert int(crc[0]) == orc.crc32c(buf) and int(crc[1]) == orc.crc32c(buf[:n // 2])
