#!/bin/bash
# tools/sq_pmc.sh OUTDIR FILTER "WORKLOAD TEXT" CMD... : the SQ counter passes of tools/attn_pmc.sh over any command (kernel names containing one of
# the comma-separated FILTER substrings are summarised) -> OUTDIR/pmc.json.  Two --pmc passes with --kernel-trace only + one --stats pass.
set -e
out=$1; filt=$2; what=$3; shift 3
root=$(pwd)
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
    -d $root/$out/p1 -o p1 --output-format csv -- "$@" > $root/$out/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_WAIT_INST_LDS \
    -d $root/$out/p2 -o p2 --output-format csv -- "$@" > $root/$out/p2.log 2>&1
rocprofv3 --kernel-trace --stats -d $root/$out/st -o st --output-format csv -- "$@" > $root/$out/st.log 2>&1
cd $root
python tools/attn_pmc_json.py $out "$filt" "$what" > $out/pmc.json
rm -rf $out/p1 $out/p2 $out/st/st_kernel_trace.csv
