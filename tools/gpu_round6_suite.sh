#!/bin/bash
export PYTHONPATH=$PWD
out=$PWD/gpurun_out
mkdir -p $out
timeout 3000 python -m pytest tests -q -m gpu > $out/r06_full_gpu_suite.txt 2>&1; echo "suite rc=$?"; tail -15 $out/r06_full_gpu_suite.txt
