#!/bin/bash
O=gpurun_out/r06
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_tail.py tests/test_gpu_scan.py tests/test_gpu_fuzz.py tests/test_gpu_multirank.py tests/test_gpu_heads.py tests/test_gpu_pipeline.py -x -q -m gpu > $O/tests15.log 2>&1; echo "tests rc=$?" >> $O/tests15.log; tail -3 $O/tests15.log
STEPS=10 python tools/r04_step_timeline.py > $O/step_timeline15.txt 2>&1; grep -v amdgpu.ids $O/step_timeline15.txt | grep -A12 "^\[aa\]"
timeout 600 python bench.py --no-config-legs > $O/bench15.json 2> $O/bench15.err; echo "bench rc=$?"
