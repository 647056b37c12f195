#!/bin/bash
# r06 measurement batch 7: r04 pair kernel on the same box, re-scoring without weight gathers (ablation), PMC passes + rocprof stats
mkdir -p gpurun_out/r06
O=gpurun_out/r06
{
  echo "== in-tree"; python tools/eval_pairs_bench.py
  echo "== r04 pair_intersect.hip (commit 08161a1) linked into today's library"; EPS_LIB_PATH=$PWD/tools/bin/libeps_pi_r04.so python tools/eval_pairs_bench.py
} > $O/eval_pairs_vs_r04.txt 2>&1
grep -v amdgpu.ids $O/eval_pairs_vs_r04.txt
{
  python tools/r04_rescore_time.py
  EPS_LIB_PATH=$PWD/tools/bin/libeps_rs_noweight.so python tools/r04_rescore_time.py
} > $O/rescore_noweight.txt 2>&1
grep -v amdgpu.ids $O/rescore_noweight.txt
tools/r06_pmc_scan.sh r06/pmc_scan > $O/pmc_scan.log 2>&1; tail -60 $O/pmc_scan.log | cut -c1-160
tools/r06_profile_bench.sh > $O/profile_bench.log 2>&1; tail -22 $O/profile_bench.log | cut -c1-260
