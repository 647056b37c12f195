#!/bin/bash
O=gpurun_out/r06
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_scan.py tests/test_gpu_fuzz.py tests/test_gpu_heads.py tests/test_gpu_tail.py tests/test_gpu_multirank.py tests/test_gpu_pipeline.py tests/test_gpu_prep.py -x -q -m gpu > $O/tests18.log 2>&1; echo "tests rc=$?" >> $O/tests18.log; tail -3 $O/tests18.log
python tools/r06_cold_timeline.py > $O/cold_timeline2.txt 2>&1; grep -v amdgpu.ids $O/cold_timeline2.txt | tail -45
timeout 600 python bench.py --no-config-legs > $O/bench18.json 2> $O/bench18.err; echo "bench rc=$?"
tools/r04_filter_cli.sh > $O/filter_cli18.txt 2>&1; grep -v amdgpu.ids $O/filter_cli18.txt | grep "scored in" | cut -c1-110
