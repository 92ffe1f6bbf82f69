#!/bin/bash
# round 6: the whole GPU suite, the bench line, the kernel trace of the headline command
tag=${1:-r06}
export PYTHONPATH=$PWD
repo=$PWD
out=$PWD/gpurun_out
mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke()" > $out/${tag}_smoke.log 2>&1; echo "smoke rc=$?"
timeout 2400 python -m pytest tests -q -m gpu -x > $out/${tag}_full_gpu_suite.txt 2>&1; echo "suite rc=$?"; tail -3 $out/${tag}_full_gpu_suite.txt
timeout 1500 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
rm -rf $out/${tag}_stats
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python3 $repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $out/${tag}_stats.log 2>&1
find $out/${tag}_stats -type f ! -name '*kernel_stats.csv' -delete
cd $repo
tail -c 400 $out/${tag}_bench.err; python3 - <<P
import json
d=json.load(open('$out/${tag}_bench.json')); print(d['value'], d['ms_per_step']); r=d['roofline']
print({k:r[k] for k in r if k not in ('measured_stream_gbs','hbm_bound_256cubed','regime','kernel','streaming_spmv_kernel','l2_counters','phases','measured_l2_gather_gbs')})
print(r['measured_l2_gather_gbs']); print(r['phases']['us_per_update'], r['phases']['spmv_phase'])
print(d['cpu_baseline'])
e=d['extra']
for k in ('c3_unstructured3d_100','c3_unstructured3d_80','c3_unstructured3d_60'):
    print(k, e[k].get('time_to_solution_ms'), e[k].get('time_to_solution_launches_ms'), {n:(e[k][n]['us_per_update']) for n in ('jacobi','ic0_multicolor_solve','ic0_solve')})
print('c2', {n:(v.get('ms'), v.get('iterations')) for n,v in e['c2_poisson2d_256'].items() if isinstance(v,dict) and 'ms' in v} if 'c2_poisson2d_256' in e else list(e.keys())[:8])
P
