# developer: SQ counters + HBM bytes of the select's kernels (tools/rs_probe.py)
export TMPDIR=/tmp; out=gpurun_out/sel; mkdir -p $out; rm -rf $out/pmc_*
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c -d $out/pmc_$c -o pmc --output-format csv -- python3 tools/rs_probe.py > $out/pmc_$c.log 2>&1 || { tail -5 $out/pmc_$c.log; exit 1; }
done
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $out/pmc_SQ -o pmc --output-format csv -- python3 tools/rs_probe.py > $out/pmc_SQ.log 2>&1 || { tail -5 $out/pmc_SQ.log; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $out/pmc_SQ2 -o pmc --output-format csv -- python3 tools/rs_probe.py > $out/pmc_SQ2.log 2>&1 || { tail -5 $out/pmc_SQ2.log; }
python3 tools/pmc_summary.py $(find $out/pmc_* -name '*counter_collection.csv') | grep rsel2 > $out/pmc_summary.txt
cat $out/pmc_summary.txt | cut -c1-200
rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_SQ $out/pmc_SQ2
