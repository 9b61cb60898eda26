#!/bin/bash
# Developer probe (round 5): is k_coadd_fused_own bound by the vector pipe?  The DEV instance with 0 / 64 / 128 extra
# independent v_fma_f32 per wave and item (ZM_FF_DBG 0 / 8 / 16), and with the pixel work off for the base line.
set -o pipefail
export TMPDIR=/tmp
for pass in 1 2; do
    ZM_FF_FORM=own timeout -k 10 300 python3 tools/ff_probe.py --dbg 0,8,16,24 2>&1 | grep ZM_FF_DBG || exit 1
done
