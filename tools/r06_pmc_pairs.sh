#!/bin/bash
# r06: fabric traffic of the generic pair kernel on the evaluation lists (tools/eval_pairs_bench.py, 2^24 pairs each): FETCH_SIZE / WRITE_SIZE
# per launch against the lists' algorithmic bytes -> the line-granularity ceiling of the uniform list.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06/pmc_pairs
mkdir -p $O
cd $R
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python3 $R/tools/eval_pairs_bench.py > $O/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, json, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob("$O/p*/")):
    for f in glob.glob(d + "*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "pair_scores_kernel" in r["Kernel_Name"] or "pair_classify" in r["Kernel_Name"]:
                agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: {"launches": len(v), "sum": sum(v), "mean": sum(v) / len(v), "max": max(v)} for c, v in cs.items()} for k, cs in agg.items()}
out["log"] = [l.strip() for l in open("$O/p1.log") if "pairs" in l]
json.dump(out, open("$O/pmc_pairs.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
PY
find $O -name "*.csv" -size +3M -delete
