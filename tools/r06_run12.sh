#!/bin/bash
O=gpurun_out/r06
mkdir -p $O
{
for E in 3000000 6060000 16777216; do
  echo "== E=$E auto"; E=$E python tools/eval_pairs_bench.py
  echo "== E=$E split"; E=$E EPS_PAIR_SHAPE=split python tools/eval_pairs_bench.py
  echo "== E=$E single"; E=$E EPS_PAIR_SHAPE=single python tools/eval_pairs_bench.py
  echo "== E=$E r04 file"; E=$E EPS_LIB_PATH=$PWD/tools/bin/libeps_pi_r04.so python tools/eval_pairs_bench.py
done
} > $O/eval_pairs_sizes.txt 2>&1
grep -v amdgpu.ids $O/eval_pairs_sizes.txt
timeout 900 python -m pytest tests/test_gpu_heads.py tests/test_gpu_pair_scores.py tests/test_gpu_scan.py -x -q -m gpu > $O/tests12.log 2>&1; echo "tests rc=$?" >> $O/tests12.log; tail -3 $O/tests12.log
