#!/bin/bash
# MFMA utilisation of the decode and GEMM kernels from counters (one PMC run, kernel-trace only): profiles/r02/mfma_pmc.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-mfma_pmc}
mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $O/p -- python3 $R/tools/decode_only.py > $O/p.log 2>&1
python3 - <<PY
import csv,glob,collections,json
cnt=collections.defaultdict(dict); dur={}; name={}
for f in glob.glob("$O/p/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "mlp_decode" in k or "gemm_f32" in k:
            d=int(r["Dispatch_Id"]); name[d]=k.split("(")[0]
            cnt[d][r["Counter_Name"]]=cnt[d].get(r["Counter_Name"],0)+float(r["Counter_Value"])
            cnt[d]["vgpr"]=r.get("VGPR_Count") or r.get("Arch_VGPR_Count"); cnt[d]["lds"]=r.get("LDS_Block_Size")
for f in glob.glob("$O/p/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if int(r["Dispatch_Id"]) in name: dur[int(r["Dispatch_Id"])]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
rows=[]
for d in sorted(name):
    c=cnt[d]; g=c.get("GRBM_GUI_ACTIVE",0)
    rows.append({"dispatch":d,"kernel":name[d],"ms":dur.get(d),**c,
                 "mfma_util": c.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/(g/8*256*4) if g else None,
                 "tflops_from_counters": c.get("SQ_INSTS_VALU_MFMA_MOPS_F32",0)*512/(dur[d]*1e-3)/1e12 if d in dur else None,
                 "shader_clock_GHz": g/8/(dur[d]*1e6) if d in dur else None})
json.dump(rows,open("$O/mfma_pmc_raw.json","w"),indent=1)
for r in rows: print(r["kernel"],r["ms"],r["mfma_util"],r["tflops_from_counters"],r["shader_clock_GHz"])
PY
