cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03z/pmc2; mkdir -p $O; cd $R
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"; do
 i=$((i+1))
 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/k$i -- python3 $R/tools/scan_one.py > $O/k$i.log 2>&1
done
python3 - <<PY
import csv,glob
for d in sorted(glob.glob("$O/k*/")):
    out={}
    for f in glob.glob(d+"*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "scan_piece" in r["Kernel_Name"]:
                out[r["Counter_Name"]]=out.get(r["Counter_Name"],0)+float(r["Counter_Value"])
    for f in glob.glob(d+"*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            if "scan_piece" in r["Kernel_Name"]:
                out.setdefault("kernel_ms",[]).append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
    print(d.split("/")[-2], out)
PY
find $O -name "*.csv" -size +3M -delete
