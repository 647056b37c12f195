#!/bin/bash
O=gpurun_out/r06
mkdir -p $O
python tools/r05_soak.py > $O/soak.txt 2>&1; grep -v amdgpu.ids $O/soak.txt | tail -5
TAIL_SORT=radix python - > $O/soak_radix.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from eps_amd import scan
scan.TAIL_SORT = "radix"
exec(open("tools/r05_soak.py").read())
PY
grep -v amdgpu.ids $O/soak_radix.txt | tail -3
EPS_TEST_POISON=1 timeout 1500 python -m pytest tests -q -m gpu > $O/gpu_suite_poisoned.txt 2>&1; echo "suite rc=$?" >> $O/gpu_suite_poisoned.txt; tail -3 $O/gpu_suite_poisoned.txt
