#!/bin/bash
export PYTHONPATH=$PWD
out=$PWD/gpurun_out
mkdir -p $out
timeout 1200 python tools/chip_trsv_probe.py u60 p3d41 p2d256 u80 q400 u100 > $out/r06b_trsv_probe.log 2>&1; echo "probe rc=$?"
cat $out/r06b_trsv_probe.log | tail -60
timeout 600 python bench.py --no-extra --no-cpu-baseline > $out/r06b_bench.json 2> $out/r06b_bench.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.load(open('gpurun_out/r06b_bench.json')); r=d['roofline']
print(d['value'], d['ms_per_step'])
print({k:r[k] for k in r if k not in ('measured_stream_gbs','hbm_bound_256cubed','regime','kernel','streaming_spmv_kernel','l2_counters')})
P
