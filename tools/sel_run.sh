export TMPDIR=/tmp; mkdir -p gpurun_out/sel
python tools/rs_probe.py > gpurun_out/sel/probe.txt 2>&1 && \
timeout -k 10 400 python -m pytest tests/test_select_bracket_gpu.py tests/test_subtract_gpu.py tests/test_device_chain_gpu.py tests/test_fullsize_gpu.py tests/test_golden_gpu.py -x -q -m gpu > gpurun_out/sel/tests.txt 2>&1
echo "tests rc $?" >> gpurun_out/sel/tests.txt
rm -rf gpurun_out/sel/prof; rocprofv3 --kernel-trace --stats -d gpurun_out/sel/prof -o rs -- python3 tools/rs_probe.py > gpurun_out/sel/prof.log 2>&1
python tools/rocpd_stats.py gpurun_out/sel/prof/rs_results.db 2>/dev/null | cut -c1-60,300- | grep rsel > gpurun_out/sel/stats.txt
python tools/rocpd_stats.py gpurun_out/sel/prof/rs_results.db 2>/dev/null | grep rsel | sed 's/"[^"]*",/K,/' >> gpurun_out/sel/stats.txt
tail -4 gpurun_out/sel/tests.txt; cat gpurun_out/sel/probe.txt | tail -1; cat gpurun_out/sel/stats.txt
