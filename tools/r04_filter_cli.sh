#!/bin/bash
# r04: the filter CLI end to end on the stand-ins, one FRESH process per run (what submit_job.py:20-21 does: one filter.py per graph);
# three runs of each -- the first pays the box's one-off file-system / code-object caches
cd $GRAFT_REPO_ROOT
W=$(mktemp -d); cd $W
for spec in "ppa adamic_ogb 4000000" "collab adamic_ogb 150000" "ddi simple 100000"; do
  set -- $spec
  for rep in 1 2 3; do
    python3 $GRAFT_REPO_ROOT/filter.py --dataset $1 --model $2 --checkpoint "$1_$2||0|$rep.pt" --synthetic --keep_top $3 2>&1 | grep -E "threshold scan|using (at least )?[0-9]+ edges|fused|dense" | sed "s/^/$1 $2 keep_top $3 run $rep: /"
  done
done
