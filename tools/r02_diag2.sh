#!/bin/bash
# r02: where does the wall time of the r01 filter stage go?  kernel + HIP API timeline of the full ppa-like filter
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/diag2
mkdir -p $O /tmp/work && cd /tmp/work
for i in 1 2 3 4 5 6 7 8; do
  python3 $R/filter.py --dataset ppa --model adamic_ogb --checkpoint "ppa_adamic_ogb||0|0.pt" --synthetic --keep_top 4000000 2>&1 | grep "scored in" >> $O/filter_walls.txt
done
for i in 1 2; do
rocprofv3 --kernel-trace --hip-trace --output-format csv -d $O/trace$i -- python3 $R/filter.py --dataset ppa --model adamic_ogb --checkpoint "ppa_adamic_ogb||0|0.pt" --synthetic --keep_top 4000000 > $O/trace$i.log 2>&1
done
ls -la $O/trace1/*/ | head
