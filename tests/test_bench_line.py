"""The stdout line of bench.py must be something the driver can parse: one compact JSON
object, the contract's keys + roofline + cpu_baseline + verify, under 4 KB whatever the run measured -- the 24 KB
record of round 3 (profiles/r03_bench_driver_flags.json) did not survive the driver's capture."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "rccl_ranks", "steps", "warmup", "ms_per_step", "higher_is_better",
                 "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "verify")


@pytest.mark.parametrize("record", ["r03_bench_driver_flags.json", "r03_bench.json", "r02_bench_driver_flags.json"])
def test_compact_line_of_a_full_record_is_small_and_complete(record, tmp_path):
    b = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles", record)))
    assert len(json.dumps(full)) > 12000  # these are the records that were too large
    extra = tmp_path / "bench_extra.json"
    text = b.emit_text(full, str(extra))
    assert "\n" not in text and len(text) <= b.LINE_TARGET_BYTES < b.LINE_HARD_LIMIT_BYTES
    line = json.loads(text)
    for k in CONTRACT_KEYS:
        assert k in line, k
    assert line["value"] == pytest.approx(full["value"], rel=1e-5)
    assert line["config"]["workload"] == full["config"]["workload"]
    r = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "alg_bytes_per_launch", "kernel_ms"):
        assert k in r, k
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-4)
    # the fraction follows from the line's own numbers: algorithmic bytes / kernel time / peak
    assert r["frac"] == pytest.approx(r["alg_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9 / r["peak"], rel=1e-4)
    c = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    # nothing is lost: the side file holds the whole record
    assert json.load(open(extra)) == full
    assert line["extra"] == "bench_extra.json"


def test_line_shrinks_before_it_fails_and_fails_above_the_hard_limit():
    b = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_driver_flags.json")))
    full["other_workloads"] = {f"w{i}" * 8: {"frames_per_s": 1.0 * i, "roofline_frac": 0.1} for i in range(400)}
    text = b.emit_text(full, "")
    assert len(text) <= b.LINE_TARGET_BYTES and "other_workloads_fps_frac" not in json.loads(text)
    full["config"] = {"workload": "x", **{f"k{i}": "v" * 90 for i in range(200)}}
    with pytest.raises(SystemExit):
        b.emit_text(full, "")


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["grid9", "1080p_80x24_truecolor"])
def test_bench_two_ranks_on_the_shared_gpu_prints_one_compact_line(workload, tmp_path):
    """`python bench.py --gpus 2` from a plain shell: bench.py becomes two ranks itself (gloo, both on the one GPU: the N > 1
    control flow -- barriers, MAX over ranks, the sharded grid with its tile exchange -- without a second device; over RCCL
    the same command refuses a one-GPU box).  One line on stdout, under the size cap, with the N > 1 fields."""
    import subprocess
    import sys

    env = dict(os.environ, ASCIICHAT_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    extra = tmp_path / "extra.json"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--workload", workload,
           "--no-cpu", "--no-d2h", "--no-hot", "--no-wire", "--others", "none", "--extra", str(extra)]
    if workload == "grid9":
        cmd += ["--batch", "9"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) <= 4096, (len(lines), [len(l) for l in lines])
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and "rccl_ranks" in line and line["scaling"] == "weak"
    if workload != "grid9":
        assert len(line["multi_gpu"]["per_rank_frames_per_s"]) == 2 and line["multi_gpu"]["backend"] == "gloo"
    assert json.load(open(extra))["n_gpus"] == 2


@pytest.mark.gpu
def test_bench_eight_ranks_on_the_shared_gpu_prints_one_compact_line(tmp_path):
    """`python bench.py --gpus 8` -- the driver's largest scaling point -- with the eight ranks on the one GPU (gloo): the
    control flow of the 8-way run (spawn, barriers, MAX over ranks, the comm.c leg over the stand-in transport) and the
    line it prints: n_gpus 8, the ranks the communicator counted, a rate per rank, under 4 KB.  NOT a scaling number."""
    import subprocess
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_comm_two_ranks import build_loopback

    env = dict(os.environ, ASCIICHAT_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", ASCIICHAT_HIP_RCCL_LIB=build_loopback())
    extra = tmp_path / "extra.json"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "5", "--warmup", "2", "--batch", "64",
           "--input-sets", "4", "--no-cpu", "--no-d2h", "--no-hot", "--no-wire", "--others", "none", "--extra", str(extra)]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) <= 4096, (len(lines), [len(l) for l in lines])
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["value"] > 0 and line["scaling"] == "weak"
    assert len(line["multi_gpu"]["per_rank_frames_per_s"]) == 8 and line["multi_gpu"]["backend"] == "gloo"
    assert line["rccl_ranks"] == 8, line.get("multi_gpu")
    assert line["config"]["global_batch"] == 8 * 64
    assert json.load(open(extra))["n_gpus"] == 8
