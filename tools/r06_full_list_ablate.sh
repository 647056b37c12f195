#!/bin/bash
# timing-only ablations of the one-pass list launch (filter_scan.hip FS_ABL_*; outputs are wrong by construction)
mkdir -p gpurun_out/r06
out=gpurun_out/r06/full_list_ablations.txt
: > $out
# (build the variants first: for a in NOULIST ...; do tools/build_variant.sh abl_$a filter_scan.hip . -DFS_ABL_$a; done)
for a in ${ABLS:-product NOULIST ULIST_NOSTORE ULIST_SMALL NOSCORESTORE NOACC}; do
  lib=""; [ "$a" != product ] && lib=tools/bin/libeps_abl_$a.so
  echo "== $a" >> $out
  EPS_LIB_PATH=$lib ABL=1 timeout 280 python tools/r06_full_list_onepass.py 2>&1 | grep -v amdgpu.ids | tail -1 >> $out
done
cat $out
