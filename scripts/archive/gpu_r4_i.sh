#!/bin/bash
# round 4, visit i: evidence on the build with the wave-level checksum closing and the shared-out small launches
O=gpurun_out/r4i; mkdir -p $O; cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
python scripts/gpu_exact_length_timing.py > $O/exact_length_timing.txt 2>&1
python scripts/gpu_exact_length_timing.py sampled_80x24_truecolor > $O/exact_length_timing_sampled.txt 2>&1
{ for p in 0 1; do echo "## ASCIICHAT_HIP_STREAM_PARTS=$p ($([ $p = 0 ] && echo 'the automatic choice' || echo 'frames never shared out: the choice before this change'))"
    ASCIICHAT_HIP_STREAM_PARTS=$p python scripts/gpu_small_batch_variants.py 2>&1 | grep -E "^#|automatic"; done; } > $O/small_batch_parts.txt 2>&1
{ echo "## one workgroup of sixteen waves (geometry 16, ASCIICHAT_HIP_STREAM_PARTS=1)"
  ASCIICHAT_HIP_STREAM_PARTS=1 TIMELINE_BATCH=1 python scripts/gpu_stream_timeline.py 1080p_80x24_truecolor -1 2>&1 | grep -v amdgpu.ids
  echo "## four workgroups of four waves (geometry 18, PARTS: the automatic choice)"
  TIMELINE_BATCH=1 python scripts/gpu_stream_timeline.py 1080p_80x24_truecolor -1 2>&1 | grep -v amdgpu.ids; } > $O/lone_frame_timeline.txt 2>&1
{ for k in 1 0; do echo "## HIP_FORCE_DEV_KERNARG=$k"; HIP_FORCE_DEV_KERNARG=$k ./scripts/ubench/launch_floor; done; } > $O/launch_floor.txt 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 --extra $O/bench_extra_driver_flags.json > $O/bench_driver_stdout.txt 2> $O/bench_driver_stderr.txt
wc -c $O/bench_driver_stdout.txt; cut -c1-400 $O/bench_driver_stdout.txt
grep "in flight 4" $O/exact_length_timing.txt
