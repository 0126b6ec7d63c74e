"""Sanitizer fuzz of the host-only C of the product (tests/fuzz/host_fuzz.c): compiled with ASan + UBSan, fed with
random / malformed strings, palettes, blobs and descriptors.  A sanitizer report aborts the binary."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_is_clean_under_sanitizers(tmp_path):
    exe = str(tmp_path / "host_fuzz")
    csrc = os.path.join(ROOT, "ascii-chat_amd", "csrc")
    subprocess.check_call(["gcc", "-std=gnu11", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-I" + os.path.join(ROOT, "include"), "-I" + csrc, os.path.join(ROOT, "tests", "fuzz", "host_fuzz.c"),
                           os.path.join(csrc, "hostutil.c"), os.path.join(csrc, "achip_host.c"), "-lm", "-lpthread", "-o", exe])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1")
    env.pop("LD_PRELOAD", None)
    out = subprocess.run([exe, "6000"], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0 and "host fuzz ok" in out.stdout, out.stderr[-3000:]
