#!/bin/bash
# round 6, visit L: the shared-out rows form as the product builds it (geometry 31, two cell slots): the GPU suite's small-launch /
# rows / drop-in tests, every form of one / eight / thirty-two small frames per launch, the drop-in's lone mono frame through bench.py
TAG=${1:-r6l}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -k "small_launches or rows or torture or dropin or graph_replay or split or multi_workgroup or tick or palettes" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 900 python3 scripts/gpu_small_run_modes.py 1 8 32 > $O/small_all_forms.txt 2>> $O/stderr.txt; cat $O/small_all_forms.txt | cut -c1-330
for rep in 1 2; do
timeout 300 python3 bench.py --others 640x480_80x24_mono --no-cpu --no-wire --no-d2h --batch-sweep '' --no-hot --extra $O/extra_$rep.json > $O/line_$rep.txt 2>> $O/stderr.txt
python3 -c "
import json; d=json.load(open('$O/extra_$rep.json')); 
for w in d.get('other_workloads', d.get('others', [])):
    print(w if isinstance(w,str) else {k:w[k] for k in w if k in ('workload','value','unit','ms_per_step','kernel_ms','kernel_variant','parts','roofline','config')})
print('headline', d['value'], d['ms_per_step'])"
done | tee $O/k1.txt
