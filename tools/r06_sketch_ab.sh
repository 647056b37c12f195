#!/bin/bash
# the production main launch (live columns, heads, pack) with hashed packed pieces vs sketch pieces (SKETCH=1): kernel ms and the
# digest of the survivor list after exact re-scoring (must be identical)
mkdir -p gpurun_out/r06
out=gpurun_out/r06/sketch_ab.txt
: > $out
for sk in 0 1 2; do
  echo "== SKETCH=$sk (2: sketch pieces on a plan with wide packed pieces)" >> $out
  WIDE=$((sk / 2)) SKETCH=$((sk > 0)) LIVE=1 PACK=1 REPS=7 timeout 280 python tools/r05_heads_ab.py 0.5 2>&1 | grep -v amdgpu.ids | grep "kernel_min_ms\|identical" >> $out
done
cat $out
