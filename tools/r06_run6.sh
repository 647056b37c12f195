#!/bin/bash
# r06 measurement batch 6: tests after the fixes, pair-kernel variants, cold timeline, bench
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_tail.py tests/test_gpu_multirank.py tests/test_gpu_pair_scores.py tests/test_gpu_scan.py -x -q -m gpu > $O/tests6.log 2>&1; echo "tests rc=$?" >> $O/tests6.log; tail -4 $O/tests6.log
{
  echo "== in-tree (PI_SMALL 128, 16-byte loads)"; python tools/eval_pairs_bench.py
  for v in pi_small0 pi_small256; do echo "== $v"; EPS_LIB_PATH=$PWD/tools/bin/libeps_$v.so python tools/eval_pairs_bench.py; done
} > $O/eval_pairs_variants2.txt 2>&1
grep -v amdgpu.ids $O/eval_pairs_variants2.txt
python tools/r06_cold_timeline.py > $O/cold_timeline.txt 2>&1; grep -v amdgpu.ids $O/cold_timeline.txt | tail -70
timeout 600 python bench.py > $O/bench6.json 2> $O/bench6.err; echo "bench rc=$?"
