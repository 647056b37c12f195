#!/bin/bash
# FETCH_SIZE / RDREQ calibration for the scan walk's access shape (tools/r05_fetch_probe.hip) -> gpurun_out/r05/fetch_probe.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05/fetch_probe
mkdir -p $O
i=0
for set in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- $R/tools/bin/r05_fetch_probe > $O/p$i.log 2>&1
done
python3 - <<PY
import csv,glob,json
out={}
for d in sorted(glob.glob("$O/p*/")):
    for f in glob.glob(d+"*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0]
            out.setdefault(k,{})
            out[k][r["Counter_Name"]]=out[k].get(r["Counter_Name"],0)+float(r["Counter_Value"])
    for f in glob.glob(d+"*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0]
            out.setdefault(k,{}).setdefault("kernel_ms",[]).append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
host=[l for l in open("$O/p1.log") if l.startswith("{")]
print(json.dumps({"host": json.loads(host[-1]) if host else None, "counters": out}, indent=1))
PY
find $O -name "*.csv" -size +1M -delete
