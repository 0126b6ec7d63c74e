"""Cross-checks the independent Python restatement of SURVEY 8(a) (tests/restatement.py) against the C oracle, byte for
byte: the torture image of SURVEY Appendix B in every (colour level, render mode) combination at three sizes, with and
without aspect + padding, the multi-byte palettes, the dithered background renderer, plus the survey's recorded
reference lengths -- which the restatement reproduces on its own, without the oracle in the loop."""
import numpy as np
import pytest

import orc
import restatement as R

TORTURE = orc.frame_torture()
COMBOS = [(cl, rm) for cl in (0, 1, 2, 3) for rm in (0, 2)]


@pytest.mark.parametrize("cl,rm", COMBOS, ids=[f"color{cl}_mode{rm}" for cl, rm in COMBOS])
def test_restatement_equals_oracle_on_the_torture_image(cl, rm):
    for (W, H) in [(80, 24), (97, 31), (200, 60)]:
        if (W, H) == (200, 60) and cl in (1,):
            continue  # 16-colour nearest-match in pure Python is slow; covered at the two smaller sizes
        assert R.convert_with_caps(TORTURE, W, H, cl, rm) == orc.convert_with_caps(TORTURE, W, H, cl, rm), (W, H)


@pytest.mark.parametrize("cl,rm", [(0, 0), (3, 0), (2, 0), (3, 2), (0, 2), (2, 2)])
def test_restatement_aspect_and_padding(cl, rm):
    for (W, H) in [(80, 24), (97, 31), (60, 40)]:
        a = R.convert_with_caps(TORTURE, W, H, cl, rm, True, True)
        assert a == orc.convert_with_caps(TORTURE, W, H, cl, rm, True, True), (W, H)
    tall = orc.frame_hash_noise(40, 200, 3)   # width-constrained and height-constrained fits, odd half-block heights
    for (W, H) in [(80, 25), (31, 77)]:
        assert R.convert_with_caps(tall, W, H, cl, rm, True, True) == orc.convert_with_caps(tall, W, H, cl, rm, True, True)


def test_restatement_reproduces_the_surveys_reference_lengths_by_itself():
    # lengths of the REFERENCE's output on the torture image at 80x24 -- no oracle involved here
    want = {(0, 0): 1159, (1, 0): 11639, (2, 0): 22429, (3, 0): 21664, (3, 2): 43121, (2, 2): 15725, (1, 2): 7697,
            (0, 2): 3921}
    for (cl, rm), n in want.items():
        assert len(R.convert_with_caps(TORTURE, 80, 24, cl, rm)) == n, (cl, rm)
    # SURVEY 8(c) anchors: length + FNV-1a-32 of the reference's output on the 640x480 gradient
    g = orc.frame_anchor_gradient()
    for (cl, rm, pad, asp), (n, h) in {(0, 0, True, True): (1721, 0x7D62F78F), (2, 0, False, False): (22255, 0xBE60A438),
                                       (3, 0, False, False): (35852, 0x885DA51D), (3, 2, False, False): (73802, 0x362719AD)}.items():
        out = R.convert_with_caps(g, 80, 24, cl, rm, pad, asp)
        assert (len(out), orc.fnv1a32(out)) == (n, h), (cl, rm)
    # Appendix B, third round: TRUECOLOR + BACKGROUND at 97x31 is the dithered 16-colour renderer, 34 602 bytes
    assert len(R.convert_with_caps(TORTURE, 97, 31, 3, 1)) == 34602
    # Appendix B: the multi-byte palettes at 97x31
    for pal in (orc.PALETTE_BLOCKS, orc.PALETTE_COOL):
        assert [len(R.convert_with_caps(TORTURE, 97, 31, cl, 0, palette=pal)) for cl in (0, 3, 2)] == [1274, 48984, 39632]
    # SURVEY A1
    assert [R.fit(*a) for a in ((1920, 1080, 80, 24), (3840, 2160, 200, 60), (3840, 2160, 400, 120), (640, 480, 80, 24),
                                (160, 96, 160, 48))] == [(80, 23), (200, 56), (400, 113), (64, 24), (160, 48)]


@pytest.mark.parametrize("palette", [orc.PALETTE_BLOCKS, orc.PALETTE_COOL, orc.PALETTE_DIGITAL, orc.PALETTE_MINIMAL, "ab",
                                     "x", "é漢😀 ."], ids=["blocks", "cool", "digital", "minimal", "ab", "x", "mixed"])
def test_restatement_palettes(palette):
    for cl in (0, 3, 2, 1):
        assert R.convert_with_caps(TORTURE, 61, 17, cl, 0, palette=palette) == \
            orc.convert_with_caps(TORTURE, 61, 17, cl, 0, palette=palette), cl
    img = orc.resize_nn(TORTURE, 61, 17)
    assert R.truecolor_bg(img, palette) == orc.print_truecolor_bg(img, palette)


def test_restatement_dither_and_other_inputs():
    for img in (orc.frame_bars(160, 120, 3), orc.frame_smooth(120, 90), orc.frame_gray(64, 48), orc.frame_hash_noise(90, 70, 2)):
        for cl, rm in ((3, 1), (3, 0), (3, 2), (0, 0)):
            assert R.convert_with_caps(img, 40, 20, cl, rm) == orc.convert_with_caps(img, 40, 20, cl, rm), (cl, rm)
    assert np.array_equal(R.resize_nearest(TORTURE, 80, 24), orc.resize_nn(TORTURE, 80, 24))
