#!/bin/bash
# r06: the round's final evidence on the frozen tree -> gpurun_out/r06/final/ (copied into profiles/r06/ afterwards)
O=gpurun_out/r06/final
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt; tail -2 $O/smoke.txt
timeout 1500 python -m pytest tests -q -m gpu > $O/gpu_suite.txt 2>&1; echo "suite rc=$?" >> $O/gpu_suite.txt; tail -3 $O/gpu_suite.txt
timeout 900 python bench.py > $O/bench_final_r06.json 2> $O/bench_final_r06.err; echo "bench rc=$?"
for kind in aa ra cn; do KIND=$kind STEPS=10 python tools/r04_step_timeline.py; done > $O/step_timeline_kinds.txt 2>&1
grep -v amdgpu.ids $O/step_timeline_kinds.txt | grep -A13 "^\[" | head -60
SEEDS=3,11 KS=4000000,150000 python tools/r03_two_kernels_same_list.py > $O/two_kernels_same_list.txt 2>&1; grep -v amdgpu.ids $O/two_kernels_same_list.txt | cut -c1-220
python tools/eval_pairs_bench.py > $O/eval_pairs.txt 2>&1; grep -v amdgpu.ids $O/eval_pairs.txt
tools/r04_filter_cli.sh > $O/filter_cli.txt 2>&1; grep -v amdgpu.ids $O/filter_cli.txt | cut -c1-200
tools/r06_profile_bench.sh > $O/profile_bench.log 2>&1; tail -14 $O/profile_bench.log | cut -c1-200
tools/r06_pmc_scan.sh r06/pmc_scan > $O/pmc_scan.log 2>&1; tail -3 $O/pmc_scan.log
python tools/r05_soak.py > $O/soak.txt 2>&1; grep -v amdgpu.ids $O/soak.txt | tail -1
