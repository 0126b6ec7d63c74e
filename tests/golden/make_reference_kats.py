#!/usr/bin/env python3
"""Extracts the known-answer values that the REFERENCE's own unit tests and headers hold for this path into
tests/golden/reference_kats.json (DATA only: inputs and expected outputs; no reference source text).

Runs in the build container only (it reads /root/reference); the committed JSON is what travels.  Sources:
  tests/unit/util/output_buffer_test.c   ob_u8 / ob_u32 decimal tables (:143-201), rep_is_profitable (:295-305),
                                         emit_rep "contains" cases (:314-327), digits_u32 (:337-351)
  tests/unit/util/ansi_fast_test.c       append_truecolor_fg/bg strings (:51-131), append_256color_* strings (:289-370),
                                         append_16color_* strings incl. the invalid-index defaults (:372-456),
                                         rgb_to_16color (:458-493), get_16color_rgb (:495-530),
                                         rgb_to_16color_dithered exact answers (:550-572: NULL buffer, edges, corner)
  tests/unit/util/palette_test.c         built-in palettes: name, UTF-8 requirement (:24-28), utf8 detection (:70-74)
  include/ascii-chat/video/ascii/palette.h   PALETTE_CHARS_* (:161-197)
  tests/unit/video/color_filter_test.c   the tint colour of every colour filter (:197-218), in color_filter_t order
  tests/unit/network/crc32_hw_test.c     CRC-32C known answers
Run:  python tests/golden/make_reference_kats.py
"""
import json
import os
import re

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def read(rel):
    return open(os.path.join(REF, rel), encoding="utf-8").read()


def main():
    out = {"_provenance": "extracted by tests/golden/make_reference_kats.py from the reference's unit tests and headers "
                          "(zfogg/ascii-chat snapshot 2026-07-23); values only"}
    ob = read("tests/unit/util/output_buffer_test.c")
    blk = ob[ob.index("ob_u8_cases[] = {"):ob.index("ParameterizedTestParameters(output_buffer, ob_u8_values)")]
    out["ob_u8"] = [[int(v), s] for v, s in re.findall(r"\{(\d+),\s*\"(\d+)\"", blk)]
    blk = ob[ob.index("ob_u32_cases[] = {"):ob.index("ParameterizedTestParameters(output_buffer, ob_u32_values)")]
    out["ob_u32"] = [[int(v.rstrip("Uu")), s] for v, s in re.findall(r"\{(\d+U?),\s*\"(\d+)\"", blk)]
    out["digits_u32"] = [[int(v.rstrip("Uu")), int(d)] for v, d in re.findall(r"cr_assert_eq\(digits_u32\((\d+U?)\),\s*(\d+)\)", ob)]
    out["rep_is_profitable"] = [[int(v), b == "true"] for v, b in
                                re.findall(r"cr_assert_eq\(rep_is_profitable\((\d+)\),\s*(true|false)\)", ob)]
    out["emit_rep_contains"] = [int(v) for v in re.findall(r"emit_rep\(&ob,\s*(\d+)\)", ob)]
    an = read("tests/unit/util/ansi_fast_test.c")
    out["rgb_to_16color"] = [[int(r), int(g), int(b), int(i)] for r, g, b, i in
                             re.findall(r"result = rgb_to_16color\((\d+),\s*(\d+),\s*(\d+)\);[^\n]*\n\s*cr_assert_eq\(result,\s*(\d+)", an)]
    # append_(16|256)color_(fg|bg)(buffer, N) ... cr_assert_str_eq(buffer, "<string>")
    out["indexed_sgr"] = [[kind, where, int(n), st.replace("\\033", "\x1b")] for kind, where, n, st in
                          re.findall(r"append_(16|256)color_(fg|bg)\(buffer,\s*(\d+)\);[^\n]*\n\s*\*result = '\\0';\s*(?://[^\n]*\n\s*)*"
                                     r"cr_assert_str_eq\(buffer,\s*\"((?:\\033)\[[\d;]+m)\"", an)]
    # get_16color_rgb(i, &r, &g, &b) followed by three cr_assert_eq(r|g|b, value)
    out["get_16color_rgb"] = [[int(i), int(r), int(g), int(b)] for i, r, g, b in
                              re.findall(r"get_16color_rgb\((\d+),[^\n]*\n\s*cr_assert_eq\(r,\s*(\d+)[^\n]*\n\s*cr_assert_eq\(g,\s*(\d+)[^\n]*\n"
                                         r"\s*cr_assert_eq\(b,\s*(\d+)", an)]
    # result = rgb_to_16color_dithered(r, g, b, x, y, w, h, NULL | error_buffer) ... cr_assert_eq(result, N)
    out["rgb_to_16color_dithered"] = [[int(r), int(g), int(b), int(x), int(y), int(w), int(h), buf != "NULL", int(v)]
                                      for r, g, b, x, y, w, h, buf, v in
                                      re.findall(r"result = rgb_to_16color_dithered\((\d+),\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+),"
                                                 r"\s*(\d+),\s*(\w+)\);[^\n]*\n\s*cr_assert_eq\(result,\s*(\d+)", an)]
    tc = re.findall(r"\{(\d+),\s*(\d+),\s*(\d+),\s*\"(\\033\[[34]8;2;[\d;]+m)\"", an)
    out["truecolor_sgr"] = [[int(r), int(g), int(b), s.replace("\\033", "\x1b")] for r, g, b, s in tc]
    cf = read("tests/unit/video/color_filter_test.c")
    out["color_filter_tints"] = [[name, int(r), int(g), int(b)] for name, r, g, b in
                                 re.findall(r"\{COLOR_FILTER_(\w+),\s*(\d+),\s*(\d+),\s*(\d+),\s*\"\w+\"\}", cf)]
    ph = read("include/ascii-chat/video/ascii/palette.h")
    out["palette_chars"] = dict(re.findall(r"#define PALETTE_CHARS_(\w+)\s+\"([^\"]*)\"", ph))
    pt = read("tests/unit/util/palette_test.c")
    blk = pt[pt.index("builtin_palette_cases[] = {"):pt.index("ParameterizedTestParameters(palette_tests, builtin_palette_tests)")]
    out["builtin_palettes"] = [[n, name, u == "true"] for n, name, u in
                               re.findall(r"\{PALETTE_(\w+),\s*\"(\w+)\",\s*PALETTE_CHARS_\w+,\s*(true|false)\}", blk)]
    blk = pt[pt.index("utf8_test_cases[] = {"):pt.index("ParameterizedTestParameters(palette_tests, utf8_encoding_tests)")]
    out["palette_requires_utf8"] = [[n, u == "true"] for n, u in re.findall(r"\{PALETTE_CHARS_(\w+),\s*\"[^\"]*\",\s*(true|false)\}", blk)]
    crc = read("tests/unit/network/crc32_hw_test.c")
    m = re.search(r"test_str = \"([^\"]*)\";.*?uint32_t expected = 0x([0-9A-Fa-f]{8});", crc, re.S)
    out["crc32c"] = [[m.group(1), int(m.group(2), 16)], ["", 0]]  # :19-20 empty data -> 0; :33-49 "Hello, World!"
    path = os.path.join(HERE, "reference_kats.json")
    json.dump(out, open(path, "w"), indent=1, ensure_ascii=False, sort_keys=True)
    print(path, {k: (len(v) if hasattr(v, "__len__") else v) for k, v in out.items() if k != "_provenance"})


if __name__ == "__main__":
    main()
