#!/bin/bash
# r06: hardware counters of ONE main launch of the production scan kernel with skipped heads (tools/r06_scan_one.py), one counter
# set per run (separate --pmc passes, --kernel-trace only: the pool's rule).  usage: r06_pmc_scan.sh <outdir>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r06/pmc_scan}
mkdir -p $O
cd $R
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python3 $R/tools/r06_scan_one.py > $O/p$i.log 2>&1
done
python3 - <<PY
import csv,glob,json
out={}
for K in ("scan_piece_kernel", "sp_refine_kernel"):
    o={}
    for d in sorted(glob.glob("$O/p*/")):
        for f in glob.glob(d+"*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if K in r["Kernel_Name"]:
                    o[r["Counter_Name"]]=o.get(r["Counter_Name"],0)+float(r["Counter_Value"])
        for f in glob.glob(d+"*/*kernel_trace.csv"):
            for r in csv.DictReader(open(f)):
                if K in r["Kernel_Name"]:
                    o.setdefault("kernel_ms",[]).append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
    out[K]=o
out["host"]=[l.strip() for l in open("$O/p1.log") if l.startswith("walked slots")]
json.dump(out,open("$O/pmc_summary.json","w"),indent=1)
print(json.dumps(out,indent=1))
PY
find $O -name "*.csv" -size +3M -delete
