#!/bin/bash
# round 6, final evidence: smoke, the whole GPU suite, PMC passes of the whole-chip kernel (memory side + L2 level), the bench line, the kernel trace
tag=${1:-r06}
export PYTHONPATH=$PWD
repo=$PWD
out=$PWD/gpurun_out
mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke()" > $out/${tag}_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $out/${tag}_smoke.log
timeout 3000 python -m pytest tests -q -m gpu > $out/${tag}_full_gpu_suite.txt 2>&1; echo "suite rc=$?"; tail -3 $out/${tag}_full_gpu_suite.txt
if [ "$2" != "nopmc" ]; then
timeout 900 bash tools/pmc_chip.sh $tag > $out/${tag}_pmc_chip.log 2>&1; echo "pmc_chip rc=$?"
timeout 1200 bash tools/pmc_chip_l2.sh $tag > $out/${tag}_pmc_chip_l2.log 2>&1; echo "pmc_chip_l2 rc=$?"
cp profiles/pmc_traffic.json profiles/${tag}_chip_traffic.md profiles/${tag}_chip_kernel_stats.csv profiles/${tag}_chip_l2_counters.md $out/ 2>/dev/null
fi
timeout 1500 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
rm -rf $out/${tag}_stats
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python3 $repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $out/${tag}_stats.log 2>&1
find $out/${tag}_stats -type f ! -name '*kernel_stats.csv' -delete
cd $repo
tail -c 300 $out/${tag}_bench.err; python3 - <<P
import json
d=json.load(open('$out/${tag}_bench.json')); print(d['value'], d['ms_per_step']); r=d['roofline']
print({k:r[k] for k in ('bound','achieved','peak','frac','frac_of_measured_ceiling','us_per_update','traffic','per_solve_time_outside_the_kernel_us')})
print(r['measured_l2_gather_gbs']); print(r['phases']['us_per_update'], r['phases']['spmv_phase']['frac'], r['phases']['gathered_bytes_alone']['frac'])
e=d['extra']
for k in ('c3_unstructured3d_100','c3_unstructured3d_80','c3_unstructured3d_60'):
    print(k, e[k].get('time_to_solution_ms'), {n:(e[k][n]['us_per_update']) for n in ('jacobi','ic0_multicolor_solve','ic0_solve')})
for k in e:
    if k.startswith('c3_mesh'): print(k, e[k]['jacobi']['us_per_update'], e[k]['jacobi_mixed_precision'])
print('c2', {n:(v.get('ms'), v.get('iterations')) for n,v in e['c2_poisson2d_256'].items() if isinstance(v,dict) and 'ms' in v})
print('c4', e.get('c4_poisson3d_256_to_solution',{}).get('jacobi'))
P
