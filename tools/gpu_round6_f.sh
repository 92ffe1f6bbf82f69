#!/bin/bash
export PYTHONPATH=$PWD
out=$PWD/gpurun_out
mkdir -p $out
timeout 1500 python tools/chip_trsv_probe.py u100 > $out/r06l_trsv_probe.log 2>&1; echo "probe rc=$?"
grep -v "amdgpu.ids" $out/r06l_trsv_probe.log | tail -20
PROBE_CHECK=0 DPCG_CHIP_TRACE=1 timeout 600 python tools/chip_trsv_probe.py u100 > $out/r06l_trace.log 2>&1; echo "rc=$?"
grep -v "amdgpu.ids" $out/r06l_trace.log | tail -32
