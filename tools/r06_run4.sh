#!/bin/bash
# r06 measurement batch 4: tail tests, re-scoring variants, main-launch A/B (generic / specialised / + column pack), timelines, bench
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_tail.py tests/test_gpu_heads.py -x -q -m gpu > $O/tests4.log 2>&1; echo "tests rc=$?" >> $O/tests4.log; tail -3 $O/tests4.log
{
  python tools/r04_rescore_time.py
  for v in rs16_16_4 rs16_4_8 rs8_8_8 rs4_4_8; do EPS_LIB_PATH=$PWD/tools/bin/libeps_$v.so python tools/r04_rescore_time.py; done
} > $O/rescore_variants.txt 2>&1
cat $O/rescore_variants.txt | grep -v amdgpu.ids
{
  echo "== generic body"; EPS_SCAN_GENERIC=1 LIVE=1 REPS=7 python tools/r05_heads_ab.py 0.5
  echo "== specialised body"; LIVE=1 REPS=7 python tools/r05_heads_ab.py 0.5
  echo "== specialised body + column pack"; PACK=1 LIVE=1 REPS=7 python tools/r05_heads_ab.py 0.5
} > $O/scan_structures_ab.txt 2>&1
grep -v amdgpu.ids $O/scan_structures_ab.txt | cut -c1-420
{
  echo "== TAIL_DEVICE=0"; TAIL_DEVICE=0 STEPS=10 python tools/r04_step_timeline.py
  echo "== library"; STEPS=10 python tools/r04_step_timeline.py
  echo "== radix"; TAIL_SORT=radix STEPS=10 python tools/r04_step_timeline.py
} > $O/step_timeline4.txt 2>&1
grep -v amdgpu.ids $O/step_timeline4.txt | grep -A14 "^\[aa\]\|^==" | head -80
timeout 600 python bench.py > $O/bench4.json 2> $O/bench4.err; echo "bench rc=$?"
