# pool throughput against hardware queues (throughput form of the factorisation)
mkdir -p gpurun_out/r03ab
export ZM_CHOL_FORM=tp
for q in 8 16 24 32; do
  for J in 8 12 16; do
    GPU_MAX_HW_QUEUES=$q python3 tools/nightly_trace.py $J 32 > gpurun_out/r03ab/q${q}_j$J.log 2>&1 || { tail -5 gpurun_out/r03ab/q${q}_j$J.log; exit 1; }
    echo "queues $q: $(grep 'J =' gpurun_out/r03ab/q${q}_j$J.log | tail -1)"
  done
done
