#!/bin/bash
# r06: hardware counters of ONE one-pass list launch (filter_scan_kernel<false, FS_EMIT>, tools/r06_full_list_one.py), one counter
# set per run (separate --pmc passes, --kernel-trace only: the pool's rule).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06/pmc_full_list
mkdir -p $O
cd $R
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python3 $R/tools/r06_full_list_one.py > $O/p$i.log 2>&1
done
python3 - <<PY
import csv,glob,json,re
o={}
for d in sorted(glob.glob("$O/p*/")):
    for f in glob.glob(d+"*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "filter_scan_kernel" in r["Kernel_Name"]:
                o[r["Counter_Name"]]=o.get(r["Counter_Name"],0)+float(r["Counter_Value"])
    for f in glob.glob(d+"*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            if "filter_scan_kernel" in r["Kernel_Name"]:
                o.setdefault("kernel_ms",[]).append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
host=[l.strip() for l in open("$O/p1.log") if l.startswith("candidates")][0]
num=lambda n: int(re.search(n+r" (\d+)",host).group(1))
cand,paths,nnz=num("candidates"),num("two-hop paths"),num("nnz")
kms=o.pop("kernel_ms"); ms=sum(kms)/len(kms)
fetch,wr=o["FETCH_SIZE"]*1024,o["WRITE_SIZE"]*1024
total=2*fetch+wr
alg=4*paths+8*cand+24*nnz
out={"command":"rocprofv3 --kernel-trace --pmc <set> --output-format csv -- python3 tools/r06_full_list_one.py  (one counter set per run: tools/r06_pmc_full_list.sh)",
     "launch":"filter_scan_kernel<false, FS_EMIT> through eps_expand_unit_list: the whole ppa-like graph in ONE launch; "+host,
     "kernel_ms_under_pmc":kms,"counters":o,
     "fabric_traffic_bytes":{"FETCH_SIZE_bytes":fetch,"WRITE_SIZE_bytes":wr,"corrected_total":total,
                             "how":"as for the scan kernel (profiles/r05/fetch_size_calibration.json): FETCH_SIZE on gfx950 tallies 128-byte requests at 64 B -> doubled"},
     "derived":{"kernel_ms_mean":ms,"algorithmic_bytes_one_pass_model":alg,"traffic_over_algorithmic_bytes":total/alg,
                "fabric_TBps":total/(ms*1e-3)/1e12,"algorithmic_TBps":alg/(ms*1e-3)/1e12,"frac_of_8TBps_on_the_model":alg/(ms*1e-3)/8e12,
                "valu_busy_of_simd_time":o["SQ_ACTIVE_INST_VALU"]/8.0/o["SQ_BUSY_CYCLES"] if o.get("SQ_BUSY_CYCLES") else None,
                "wave_issue_share":o["SQ_ACTIVE_INST_ANY"]/o["SQ_WAVE_CYCLES"],"waves_waiting_share":o["SQ_WAIT_ANY"]/o["SQ_WAVE_CYCLES"],
                "lds_bank_conflict_share_of_lds_cycles":o["SQ_LDS_BANK_CONFLICT"]/o["SQ_LDS_IDX_ACTIVE"],
                "l2_hit_rate":o["TCC_HIT_sum"]/(o["TCC_HIT_sum"]+o["TCC_MISS_sum"]),
                "valu_lane_instructions_per_two_hop_path":o["SQ_INSTS_VALU"]*64/paths,
                "wave_instructions_per_two_hop_path":{k:o[k]/paths for k in ("SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_LDS","SQ_INSTS_VMEM_WR","SQ_INSTS_VMEM_RD")}}}
json.dump(out,open("$O/full_list_pmc.json","w"),indent=1)
print(json.dumps(out["derived"],indent=1))
PY
find $O -name "*.csv" -size +3M -delete
