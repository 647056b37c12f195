#!/bin/bash
# r06 measurement batch 5: tail + multirank tests, pair-kernel variants on the evaluation lists, bench
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_tail.py tests/test_gpu_multirank.py tests/test_gpu_pair_scores.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/tests5.log 2>&1; echo "tests rc=$?" >> $O/tests5.log; tail -4 $O/tests5.log
{
  echo "== in-tree (PI_SMALL 256, 16-byte loads)"; python tools/eval_pairs_bench.py
  for v in pi_small64 pi_small128; do echo "== $v"; EPS_LIB_PATH=$PWD/tools/bin/libeps_$v.so python tools/eval_pairs_bench.py; done
} > $O/eval_pairs_variants.txt 2>&1
grep -v amdgpu.ids $O/eval_pairs_variants.txt
timeout 600 python bench.py > $O/bench5.json 2> $O/bench5.err; echo "bench rc=$?"
