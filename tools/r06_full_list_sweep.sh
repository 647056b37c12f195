#!/bin/bash
# geometry sweep of the list kernels (filter_scan.hip): tile size (variant libraries) x number of id windows
mkdir -p gpurun_out/r06
out=gpurun_out/r06/full_list_sweep.txt
: > $out
for lib in "" tools/bin/libeps_tb13.so tools/bin/libeps_tb14.so; do
  for w in 1 3 5 9 18; do
    EPS_LIB_PATH=$lib EPS_FS_MIN_WIN=$w QUICK=1 timeout 300 python tools/r06_full_list_split.py 2>&1 | grep -v amdgpu.ids >> $out
  done
done
cat $out
