"""Builds the product's host C against a mock HIP runtime (tests/mockhip): device memory is host memory, launches run the
product kernels under the CPU fiber emulator.  TESTS ONLY -- lets the CPU suite drive the drop-in layer (direct path and the
flat-combining layer), plans and the frame table, which otherwise need a GPU box.  The product library is not involved and
has no such mode (it fails with ERR_NO_DEVICE without a GPU)."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ascii-chat_amd", "csrc")
INC = os.path.join(ROOT, "include")
MOCK = os.path.join(ROOT, "tests", "mockhip")
EMU = os.path.join(ROOT, "tests", "hipemu")
BUILD = os.path.join(MOCK, "_build")
HOST_C = ["dropin.c", "combine.c", "plan.c", "achip_host.c", "hostutil.c", "buffer_pool.c", "frame_table.c", "frame_dense.c", "comm.c"]
HIP_INC = "/opt/rocm/include"


def _newer(target, deps):
    return os.path.exists(target) and all(os.path.getmtime(d) <= os.path.getmtime(target) for d in deps)


def _deps():
    d = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(INC, f) for f in os.listdir(INC)]
    d += [os.path.join(MOCK, f) for f in os.listdir(MOCK) if f.endswith((".c", ".cpp"))]
    d += [os.path.join(EMU, f) for f in os.listdir(EMU) if f.endswith((".cpp", ".h", ".hpp"))]
    return d


def build_library():
    """-> path of libasciichat_mock.so: host C + mock HIP + launches on the emulator"""
    so = os.path.join(BUILD, "libasciichat_mock.so")
    if _newer(so, _deps()):
        return so
    os.makedirs(BUILD, exist_ok=True)
    objs = []
    tag = ".%d" % os.getpid()  # objects and library of this process only, then one rename: concurrent test processes (pytest -n)
    for src in [os.path.join(CSRC, f) for f in HOST_C] + [os.path.join(MOCK, "mock_hip.c")]:  # never link or load each other's halves
        obj = os.path.join(BUILD, os.path.basename(src).replace(".c", tag + ".o"))
        subprocess.check_call(["gcc", "-std=gnu11", "-O1", "-g", "-fPIC", "-pthread", "-I" + INC, "-I" + CSRC, "-I" + HIP_INC,
                               "-c", src, "-o", obj])
        objs.append(obj)
    lobj = os.path.join(BUILD, "mock_launch" + tag + ".o")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-DACHIP_ALL_GEOMETRIES", "-I" + EMU, "-I" + CSRC, "-I" + INC, "-c",
                           os.path.join(MOCK, "mock_launch.cpp"), "-o", lobj])
    subprocess.check_call(["g++", "-shared", "-o", so + tag, *objs, lobj, "-lpthread", "-lm", "-ldl"])
    os.replace(so + tag, so)
    for o in objs + [lobj]:
        os.remove(o)
    return so


def build_thread_harness(sanitizer=None, harness="dropin_threads_mock"):
    """-> path of a plain-C harness (tests/mockhip/<harness>.c) linked with the host C, the mock runtime and the arithmetic
    stand-in for the kernels (mock_launch_simple.c: no fibers, so ThreadSanitizer can follow everything)"""
    tag = sanitizer or "plain"
    exe = os.path.join(BUILD, f"{harness}_{tag.replace(',', '_')}")
    if _newer(exe, _deps()):
        return exe
    os.makedirs(BUILD, exist_ok=True)
    flags = ["-std=gnu11", "-O1", "-g", "-pthread", "-DACHIP_ALL_GEOMETRIES", "-I" + INC, "-I" + CSRC, "-I" + HIP_INC]
    if sanitizer:
        flags += ["-fsanitize=" + sanitizer, "-fno-omit-frame-pointer", "-Wno-tsan"]
    srcs = [os.path.join(CSRC, f) for f in HOST_C] + [os.path.join(MOCK, f) for f in
                                                        ("mock_hip.c", "mock_launch_simple.c", harness + ".c")]
    subprocess.check_call(["gcc", *flags, *srcs, "-o", exe, "-lm", "-ldl"])
    return exe


_pkg = None


def package():
    """the package's binding module bound to the MOCK library (a private copy of the module: the real one stays untouched)"""
    global _pkg
    if _pkg is None:
        import importlib.util
        spec = importlib.util.spec_from_file_location("achip_binding_mock", os.path.join(ROOT, "ascii-chat_amd", "binding.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod._lib = mod._bind(C.CDLL(build_library()))
        _pkg = mod
    return _pkg
