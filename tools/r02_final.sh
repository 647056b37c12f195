#!/bin/bash
# r02 final evidence: GPU test suite, bench line, rocprofv3 kernel statistics of the bench command, PMC passes of the dominant kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/final_${1:-a}
mkdir -p $O /tmp/fw
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
(cd /tmp/fw && for i in 1 2 3; do python $GRAFT_REPO_ROOT/filter.py --dataset ppa --model adamic_ogb --checkpoint "ppa_adamic_ogb||0|0.pt" --synthetic --keep_top 4000000 2>&1 | grep -E "scored in|threshold scan"; done) > $O/filter_cli.txt 2>&1
bash tools/r02_profile_bench.sh final_${1:-a}/prof > $O/prof.log 2>&1
python tools/make_scan_stamps.py > /dev/null 2>&1 && timeout 300 python tools/scan_stamps.py > $O/stamps.txt 2>&1
cat $O/pytest.txt $O/filter_cli.txt; head -c 700 $O/bench.json; tail -5 $O/stamps.txt
