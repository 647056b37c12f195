#!/bin/bash
# PMC passes over one main launch of eps_filter_scan (each counter set in its own run, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-pmc_scan}
mkdir -p $O
python3 $R/tools/scan_stamps.py > $O/stamps.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python3 $R/tools/scan_one.py > $O/p$i.log 2>&1
done
python3 - <<PY
import csv,glob,collections,json
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
json.dump(out,open("$O/summary.json","w"),indent=1)
print(json.dumps(out,indent=1))
PY
find $O -name "*.csv" -size +5M -delete
cat $O/stamps.txt
