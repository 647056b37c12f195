#!/bin/bash
# timing-only: the production main launch (live columns, heads, pack) with the packed pieces' hash insert replaced by ONE
# non-returning LDS add per path (scan_pieces.hip SP_ABL_PACKED_NOHASH; build: tools/build_variant.sh nohash scan_pieces.hip . -DSP_ABL_PACKED_NOHASH)
# against the product -- the most ANY table structure for the sparse tail could save.
mkdir -p gpurun_out/r06
out=gpurun_out/r06/packed_nohash.txt
: > $out
for lib in "" tools/bin/libeps_nohash.so; do
  echo "== ${lib:-product}" >> $out
  EPS_LIB_PATH=$lib SKETCH=0 LIVE=1 PACK=1 REPS=7 timeout 280 python tools/r05_heads_ab.py 0.5 2>&1 | grep -v amdgpu.ids | grep kernel_min_ms >> $out
done
cat $out | cut -c1-400
