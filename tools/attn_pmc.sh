#!/bin/bash
# tools/attn_pmc.sh OUTDIR : two rocprofv3 --pmc passes (8 SQ counters each, --kernel-trace only) + one --stats pass over
# tools/attn_pmc_driver.py; summarised by tools/attn_pmc_json.py into OUTDIR/attn_pmc.json.  Run on the GPU box.
set -e
out=${1:-gpurun_out/attn_pmc}
root=$(pwd)
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
    -d $root/$out/p1 -o p1 --output-format csv -- python $root/tools/attn_pmc_driver.py > $root/$out/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_WAIT_INST_LDS \
    -d $root/$out/p2 -o p2 --output-format csv -- python $root/tools/attn_pmc_driver.py > $root/$out/p2.log 2>&1
rocprofv3 --kernel-trace --stats -d $root/$out/st -o st --output-format csv -- python $root/tools/attn_pmc_driver.py > $root/$out/st.log 2>&1
cd $root
python tools/attn_pmc_json.py $out > $out/attn_pmc.json
