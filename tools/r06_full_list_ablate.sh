#!/bin/bash
# timing-only ablations of the one-pass list launch (filter_scan.hip FS_ABL_*; outputs are wrong by construction)
mkdir -p gpurun_out/r06
out=gpurun_out/r06/full_list_ablations.txt
: > $out
for a in ${ABLS:-"" NOULIST NOSCORESTORE NOACC NORECSTORE}; do
  lib=""; [ -n "$a" ] && lib=tools/bin/libeps_abl_$a.so
  echo "== ${a:-product}" >> $out
  EPS_LIB_PATH=$lib ABL=1 timeout 280 python tools/r06_full_list_onepass.py 2>&1 | grep -v amdgpu.ids | tail -1 >> $out
done
cat $out
