#!/bin/bash
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 600 python -m pytest tests/test_gpu_pair_scores.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/tests10.log 2>&1; echo "tests rc=$?" >> $O/tests10.log; tail -3 $O/tests10.log
{
  echo "== in-tree (device-side choice: split with PI_SMALL 128, PI_TICKET 4 -- or one pair at a time)"; python tools/eval_pairs_bench.py
  for v in pi_r04; do echo "== $v"; EPS_LIB_PATH=$PWD/tools/bin/libeps_$v.so python tools/eval_pairs_bench.py; done
} > $O/eval_pairs_choice.txt 2>&1
grep -v amdgpu.ids $O/eval_pairs_choice.txt
timeout 600 python bench.py > $O/bench10.json 2> $O/bench10.err; echo "bench rc=$?"
