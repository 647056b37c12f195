#!/bin/bash
# Build tools/bin/libeps_<name>.so: the whole library with csrc/scan_pieces.hip taken from <source> (a file, or a git
# revision like HEAD~1) and extra -D flags -- for same-box A/B runs through EPS_LIB_PATH (tools/r04_scan_ab.py).
# usage: build_lib_variant.sh <name> <file-or-rev> [-DFLAG ...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/edge-proposal-sets_amd/csrc
name=$1; src=$2; shift 2
tmp=$CSRC/_sp_var_$name.hip
if [ -f "$src" ]; then cp "$src" "$tmp"; else git -C "$ROOT" show "$src:edge-proposal-sets_amd/csrc/scan_pieces.hip" > "$tmp"; fi
make -C "$CSRC" -s -j8
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-result "$@" -c "$tmp" -o /tmp/_sp_var_$name.o
rm -f "$tmp"
objs=$(ls $CSRC/build/*.o | grep -v scan_pieces.o)
mkdir -p $ROOT/tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/bin/libeps_$name.so /tmp/_sp_var_$name.o $objs
echo built tools/bin/libeps_$name.so
