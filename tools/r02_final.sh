#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/final_${1:-a}
mkdir -p $O /tmp/fw
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
(cd /tmp/fw && for i in 1 2 3; do python $GRAFT_REPO_ROOT/filter.py --dataset ppa --model adamic_ogb --checkpoint "ppa_adamic_ogb||0|0.pt" --synthetic --keep_top 4000000 2>&1 | grep -E "scored in|threshold scan"; done) > $O/filter_cli.txt 2>&1
timeout 300 python tools/spmm_reorder.py > $O/spmm_reorder.txt 2>&1
timeout 300 python tools/scan_bench.py --reps 3 2>&1 | grep -v amdgpu > $O/scan_bench.txt
cat $O/pytest.txt $O/filter_cli.txt $O/spmm_reorder.txt $O/scan_bench.txt; head -c 600 $O/bench.json
