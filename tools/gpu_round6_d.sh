#!/bin/bash
export PYTHONPATH=$PWD
out=$PWD/gpurun_out
mkdir -p $out
PROBE_CHECK=0 DPCG_CHIP_TRACE=1 timeout 600 python tools/chip_trsv_probe.py u80 u100 p3d41 > $out/r06g_trace.log 2>&1; echo "rc=$?"
grep -v "amdgpu.ids" $out/r06g_trace.log | tail -120
