#!/bin/bash
# Developer tool: counter passes of the fused coadd kernel on the probe stack (tools/ff_probe.py), for each form.
#   bash tools/ff_pmc.sh <tag> ["dma own"]     -> gpurun_out/<tag>/pmc_summary_<form>.txt
set -o pipefail
tag=${1:-ffpmc}
forms=${2:-"dma own"}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
P="tools/ff_probe.py --reps 2 --dbg 0"
for form in $forms; do
    export ZM_FF_FORM=$form
    for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS" \
             "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT" \
             "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM"; do
        n=$(echo $c | tr ' ' '_' | cut -c1-40)
        timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d $out/pmc_${form}_$n -o pmc --output-format csv -- python3 $P > $out/pmc_${form}_$n.log 2>&1 || { tail -5 $out/pmc_${form}_$n.log; }
    done
    python3 tools/pmc_summary.py $(find $out/pmc_${form}_* -name '*counter_collection.csv') > $out/pmc_summary_$form.txt
    echo "== $form"
    grep -E "k_coadd_fused" $out/pmc_summary_$form.txt
    rm -rf $out/pmc_${form}_*/
done
