#!/bin/bash
# r02 diagnostics on the r01 kernels: counter list, bimodality of the scored expansion, LDS counters of one launch
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/diag1
mkdir -p $O
rocprofv3 -L > $O/counters_list.txt 2>&1
rocm-smi --showclocks > $O/clocks_before.txt 2>&1
for i in 1 2 3 4 5 6; do
  MAX_PATHS=$((1<<29)) python3 $R/tools/expand_blocks.py 2>&1 | tail -1 >> $O/bimodal.txt
done
rocm-smi --showclocks > $O/clocks_after.txt 2>&1
cd $O
REPS=1 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS --output-format csv -d $O/pmc1 -- python3 $R/tools/expand_one_block.py > $O/pmc1.log 2>&1
REPS=1 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc2 -- python3 $R/tools/expand_one_block.py > $O/pmc2.log 2>&1
REPS=1 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc3 -- python3 $R/tools/expand_one_block.py > $O/pmc3.log 2>&1
# keep only the counter csvs (small)
find $O -name "*.csv" -size +20M -delete
ls -R $O | head -50
