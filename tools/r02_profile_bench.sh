#!/bin/bash
# r02: rocprofv3 kernel statistics of the bench command + fabric-traffic counters of the dominant kernel's launch
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-prof_bench}
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python3 $R/tools/scan_one.py > $O/p$i.log 2>&1
done
python3 - <<PY
import csv,glob,json
out={}
for d in sorted(glob.glob("$O/p*/")):
    for f in glob.glob(d+"*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "filter_scan_kernel" in r["Kernel_Name"]:
                out[r["Counter_Name"]]=out.get(r["Counter_Name"],0)+float(r["Counter_Value"])
    for f in glob.glob(d+"*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            if "filter_scan_kernel" in r["Kernel_Name"]:
                out.setdefault("kernel_ms",[]).append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
json.dump(out,open("$O/pmc_summary.json","w"),indent=1)
print(json.dumps(out,indent=1))
PY
cp $O/stats/*/*kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null
find $O -name "*.csv" -size +3M -delete
head -12 $O/bench_kernel_stats.csv
