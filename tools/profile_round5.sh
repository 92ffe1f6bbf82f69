#!/bin/bash
# Round-5 evidence on the GPU box:  gpurun --timeout 2400 -- 'bash tools/profile_round5.sh r05'
# the bench line, the kernel trace of the headline command, the whole-chip solve's kernel trace + PMC traffic (tools/pmc_chip.sh)
tag=${1:-r05}
export PYTHONPATH=$PWD
repo=$PWD
out=$PWD/gpurun_out
mkdir -p $out
bash tools/pmc_chip.sh $tag > $out/${tag}_pmc_chip.log 2>&1
cp profiles/pmc_traffic.json $out/${tag}_pmc_traffic.json
cp profiles/${tag}_chip_traffic.md $out/ 2>/dev/null
cp profiles/${tag}_chip_kernel_stats.csv $out/ 2>/dev/null
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf $out/${tag}_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python3 $repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $out/${tag}_stats.log 2>&1
find $out/${tag}_stats -type f ! -name '*kernel_stats.csv' -delete
tail -c 600 $out/${tag}_bench.err; python3 -c "
import json; d=json.load(open('$out/${tag}_bench.json')); print(d['value'], d['ms_per_step']); r=d['roofline']; print({k:r[k] for k in r if k not in ('measured_stream_gbs','hbm_bound_256cubed','regime','kernel','streaming_spmv_kernel')}); print(r['hbm_bound_256cubed']['frac'], r['hbm_bound_256cubed']['in_loop']); print(d['cpu_baseline'])"
tail -20 $out/${tag}_pmc_chip.log
