#!/bin/bash
set -o pipefail
out=gpurun_out/${1:-r04g}
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_object_route_gpu.py tests/test_scripts_gpu.py tests/test_object_api_gpu.py tests/test_device_chain_gpu.py tests/test_detect_gpu.py -m gpu -x -q > $out/tests.log 2>&1 || { tail -60 $out/tests.log; exit 1; }
tail -3 $out/tests.log
for r in host device; do
ZM_OBJECT_API=$r timeout -k 10 600 python3 tools/object_api_fullsize.py 8 > $out/objapi_$r.log 2>&1 || { tail -30 $out/objapi_$r.log; exit 1; }
grep -E "from_images|frames written" $out/objapi_$r.log
done
