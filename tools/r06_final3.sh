#!/bin/bash
# r06, last state: the whole GPU suite with the allocator's free blocks poisoned, then the bench line -> gpurun_out/r06/final3/
O=gpurun_out/r06/final3
mkdir -p $O
EPS_TEST_POISON=1 timeout 2400 python -m pytest tests -q -m gpu > $O/gpu_suite_poisoned.txt 2>&1; echo "suite rc=$?" >> $O/gpu_suite_poisoned.txt; tail -3 $O/gpu_suite_poisoned.txt
timeout 900 python bench.py > $O/bench_final_r06.json 2> $O/bench_final_r06.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$O/bench_final_r06.json"))
print({k:d[k] for k in ("value","ms_per_step","cold_ms_per_step","sustained_ms_per_step")}, d["roofline"]["frac"], d["roofline"]["kernel_ms"])
PY
