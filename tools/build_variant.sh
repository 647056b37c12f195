#!/bin/bash
# Build tools/bin/libeps_<name>.so: the whole library with ONE csrc unit rebuilt with extra -D flags (or taken from a git
# revision) -- for same-box A/B runs through EPS_LIB_PATH.
# usage: build_variant.sh <name> <unit.hip> <file-or-rev-or-'.'> [-DFLAG ...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/edge-proposal-sets_amd/csrc
name=$1; unit=$2; src=$3; shift 3
tmp=$CSRC/_var_$name.hip
if [ "$src" = "." ]; then cp "$CSRC/$unit" "$tmp"; elif [ -f "$src" ]; then cp "$src" "$tmp"; else git -C "$ROOT" show "$src:edge-proposal-sets_amd/csrc/$unit" > "$tmp"; fi
make -C "$CSRC" -s -j8
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-result "$@" -c "$tmp" -o /tmp/_var_$name.o
rm -f "$tmp"
objs=$(ls $CSRC/build/*.o | grep -v "/${unit%.hip}.o")
mkdir -p $ROOT/tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/bin/libeps_$name.so /tmp/_var_$name.o $objs
echo built tools/bin/libeps_$name.so
