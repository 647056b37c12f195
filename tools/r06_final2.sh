#!/bin/bash
# r06, after the one-pass list and the RCCL one-rank test: smoke, the whole GPU suite, the bench line, the two-kernels list comparison
# (eps_filter_scan's tile plan changed) and the CLI stand-ins -> gpurun_out/r06/final2/
O=gpurun_out/r06/final2
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt; tail -2 $O/smoke.txt
timeout 1800 python -m pytest tests -q -m gpu > $O/gpu_suite.txt 2>&1; echo "suite rc=$?" >> $O/gpu_suite.txt; tail -3 $O/gpu_suite.txt
timeout 900 python bench.py > $O/bench_final_r06.json 2> $O/bench_final_r06.err; echo "bench rc=$?"
SEEDS=3,11 KS=4000000,150000 python tools/r03_two_kernels_same_list.py > $O/two_kernels_same_list.txt 2>&1; grep -v amdgpu.ids $O/two_kernels_same_list.txt | cut -c1-220
tools/r04_filter_cli.sh > $O/filter_cli.txt 2>&1; grep -v amdgpu.ids $O/filter_cli.txt | cut -c1-200
