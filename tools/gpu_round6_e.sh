#!/bin/bash
export PYTHONPATH=$PWD
out=$PWD/gpurun_out
mkdir -p $out
timeout 1200 python tools/chip_trsv_probe.py u60 p3d41 p2d256 u80 q400 > $out/r06j_trsv_probe.log 2>&1; echo "probe rc=$?"
grep -v "amdgpu.ids" $out/r06j_trsv_probe.log | tail -60
PROBE_CHECK=0 DPCG_CHIP_TRACE=1 timeout 600 python tools/chip_trsv_probe.py u80 > $out/r06j_trace.log 2>&1; echo "rc=$?"
grep -v "amdgpu.ids" $out/r06j_trace.log | tail -40
