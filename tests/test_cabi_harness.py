"""tests/cabi/ascii_test_port.c: a plain-C caller that includes only include/asciichat_render.h, links the .so and
replays the assertions of the reference's tests/unit/video/ascii_test.c at the drop-in boundary."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cabi", "ascii_test_port.c")
EXE = os.path.join(ROOT, "tests", "cabi", "ascii_test_port")
LIBDIR = os.path.join(ROOT, "ascii-chat_amd")


def build():
    subprocess.check_call(["gcc", "-std=gnu11", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           SRC, "-o", EXE, "-L" + LIBDIR, "-lasciichat_hip", "-Wl,-rpath," + LIBDIR])
    return EXE


def test_harness_compiles_against_the_public_header_only():
    assert os.path.exists(os.path.join(LIBDIR, "libasciichat_hip.so")), "build the library first (__graft_entry__.build)"
    build()
    # without a GPU the library must fail loudly, not fall back: the harness exits non-zero and says why
    import ctypes
    have_gpu = ctypes.CDLL(os.path.join(LIBDIR, "libasciichat_hip.so")).asciichat_hip_device_count() > 0
    if not have_gpu:
        r = subprocess.run([EXE], capture_output=True, text=True)
        assert r.returncode != 0 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_reference_unit_test_assertions_through_the_c_abi():
    r = subprocess.run([build()], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().splitlines()[-1].startswith("ok:")


# ---- buffer_pool_* : the invariants of the reference's tests/unit/util/buffer_pool_test.c + this library's additions ----
POOL_SRC = os.path.join(ROOT, "tests", "cabi", "buffer_pool_test_port.c")
POOL_EXE = os.path.join(ROOT, "tests", "cabi", "buffer_pool_test_port")


def build_pool():
    subprocess.check_call(["gcc", "-std=gnu11", "-O1", "-Wall", "-Wextra", "-Werror", "-pthread",
                           "-I" + os.path.join(ROOT, "include"), POOL_SRC, "-o", POOL_EXE, "-L" + LIBDIR,
                           "-lasciichat_hip", "-Wl,-rpath," + LIBDIR])
    return POOL_EXE


def test_buffer_pool_invariants_host_blocks():
    """Runs everywhere: without a GPU the frame class hands out plain host blocks (the reference's malloc fallback)."""
    import ctypes
    have_gpu = ctypes.CDLL(os.path.join(LIBDIR, "libasciichat_hip.so")).asciichat_hip_device_count() > 0
    r = subprocess.run([build_pool()] + (["gpu"] if have_gpu else []), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().splitlines()[-1].startswith("ok:")


@pytest.mark.gpu
def test_buffer_pool_invariants_with_pinned_frames():
    r = subprocess.run([build_pool(), "gpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "with the pinned frame class" in r.stdout


@pytest.mark.parametrize("san", ["thread", "address,undefined"])
def test_buffer_pool_under_sanitizers(san, tmp_path):
    """The pool's source compiled together with the harness under TSan / ASan + UBSan (host blocks only: the lock-free
    small-object list, statistics, shrink, eight threads)."""
    exe = str(tmp_path / "pool_san")
    cmd = ["gcc", "-std=gnu11", "-O1", "-g", "-fsanitize=" + san, "-pthread", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(LIBDIR, "csrc"), "-I/opt/rocm/include", POOL_SRC, os.path.join(LIBDIR, "csrc", "buffer_pool.c"),
           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True)
    if b.returncode != 0:
        pytest.skip("sanitizer build not available here: " + b.stderr[-300:])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", TSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "ok:" in r.stdout and "Sanitizer" not in r.stderr, r.stdout[-500:] + r.stderr[-3000:]


# ---- the batch layer from plain C: one server tick (frame table -> plan -> render + wire stage -> host-side verification) ----
TICK_SRC = os.path.join(ROOT, "tests", "cabi", "server_tick_port.c")
TICK_EXE = os.path.join(ROOT, "tests", "cabi", "server_tick_port")


def build_tick():
    subprocess.check_call(["gcc", "-std=gnu11", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           "-I/opt/rocm/include", TICK_SRC, "-o", TICK_EXE, "-L" + LIBDIR, "-lasciichat_hip", "-L/opt/rocm/lib",
                           "-lamdhip64", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"])
    return TICK_EXE


def test_server_tick_harness_compiles_in_plain_c():
    build_tick()


@pytest.mark.gpu
def test_server_tick_in_plain_c():
    """No Python in the data path: a C program publishes twelve clients' frames, renders them with one plan (once as row
    bands + the stand-alone wire kernel, once as whole frames with the fused CRC / headers), copies the packets back and
    verifies frames (== the drop-in entry point's output), header fields and both CRC-32Cs with its own bit-serial CRC."""
    r = subprocess.run([build_tick()], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().splitlines()[-1].startswith("ok:") and "fused CRC yes" in r.stdout
    # round 4: the same frames from the images the targets sample (stage x N + commit, then the one-call batch), the send side
    # (exact-length frames + wire stage) through plan_render_packets_packed
    assert "tick 3 (stage x N + commit" in r.stdout and "tick 4 (publish_sampled_batch" in r.stdout
    assert "grid tick: 9 sources" in r.stdout  # the 3x3 grid with its tiles through the library's RCCL layer (world of one)
