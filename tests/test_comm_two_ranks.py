"""ascii-chat_amd/csrc/comm.c with a world of TWO ranks (VERDICT r2 "next" 1; gap 5).

Two forms of the same worker (tests/comm_worker.py), both through the C-ABI:
  * over a stand-in transport (tests/cabi/loopback_rccl.c, selected with ASCIICHAT_HIP_RCCL_LIB) with both ranks on ONE
    GPU -- RCCL refuses two ranks on a device, and every box this suite has seen so far has one.  This covers what
    comm.c decides itself: in-place offsets of the slab and packed gathers, lengths-first sizing, tile slots of unevenly
    sharded sources;
  * over the real librccl, one rank per GPU -- skipped unless hipGetDeviceCount() >= 2.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOOP_SRC = os.path.join(ROOT, "tests", "cabi", "loopback_rccl.c")
LOOP_SO = os.path.join(ROOT, "tests", "cabi", "libloopback_rccl.so")
WORKER = os.path.join(ROOT, "tests", "comm_worker.py")


def build_loopback():
    subprocess.check_call(["gcc", "-std=gnu11", "-O1", "-Wall", "-Wextra", "-Werror", "-shared", "-fPIC", "-I/opt/rocm/include",
                           LOOP_SRC, "-o", LOOP_SO, "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-Wl,-rpath,/opt/rocm/lib"])
    return LOOP_SO


def test_loopback_transport_builds_and_exports_what_comm_c_resolves():
    import ctypes
    so = ctypes.CDLL(build_loopback())
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclCommCount", "ncclAllGather", "ncclGroupStart",
                 "ncclGroupEnd", "ncclGetErrorString"):
        assert hasattr(so, name), name


def run_world(world, env_extra, shared_gpu, tmp_path, full=False):
    uid = str(tmp_path / "uid.bin")
    env = dict(os.environ, **env_extra)
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), uid, "1" if shared_gpu else "0"] + (["full"] if full else []), env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and out.strip().splitlines()[-1] == f"ok rank {r}", f"rank {r}:\n{out[-3000:]}"


@pytest.mark.gpu
def test_comm_world2_over_loopback_transport_on_one_gpu(tmp_path):
    run_world(2, {"ASCIICHAT_HIP_RCCL_LIB": build_loopback()}, True, tmp_path)


@pytest.mark.gpu
def test_comm_world8_over_loopback_transport_on_one_gpu(tmp_path):
    """The driver's 8-GPU run, de-risked without the hardware: comm.c at world 8 over the stand-in
    transport, eight processes on the one GPU -- BASELINE's 256 frames sharded 32 x 8 through the slab and packed gathers,
    frames of configs[4]'s size (1.8 MB) through the packed gather, configs[3]'s nine sources dealt (2, 1, 1, 1, 1, 1, 1, 1)
    through grid_slot_of (and five / three sources: ranks that own nothing).  No scaling number follows from this."""
    run_world(8, {"ASCIICHAT_HIP_RCCL_LIB": build_loopback()}, True, tmp_path, full=True)


@pytest.mark.gpu
def test_comm_world2_over_rccl(tmp_path):
    import torch  # (not the library's own count: loading it ahead of torch would bring a second HIP runtime into this process)
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device); the loopback test covers comm.c's own logic")
    run_world(2, {}, False, tmp_path)
