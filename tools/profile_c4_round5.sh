#!/bin/bash
# The 256^3 PCG loop (BASELINE config 4's system) under the profiler, round 5:  gpurun --timeout 1500 -- 'bash tools/profile_c4_round5.sh r05'
# kernel trace + stats of tools/trace_run_c4.py, its per-kernel medians / gaps (tools/trace_gaps.py), and the in-loop probe
tag=${1:-r05}
export PYTHONPATH=$PWD
repo=$PWD
out=$PWD/gpurun_out
mkdir -p $out
python tools/c4_inloop_probe.py > $out/${tag}_c4_inloop_probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $out/${tag}_stats_c4
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats_c4 -- python3 $repo/tools/trace_run_c4.py > $out/${tag}_stats_c4.log 2>&1
cd $repo
tr=$(ls $out/${tag}_stats_c4/*/*kernel_trace.csv | head -1)
python3 tools/trace_gaps.py $tr > $out/${tag}_kernel_trace_256cubed_summary.txt 2>&1
st=$(ls $out/${tag}_stats_c4/*/*kernel_stats.csv | head -1)
cp $st $out/${tag}_kernel_stats_256cubed.csv
find $out/${tag}_stats_c4 -type f -delete
cat $out/${tag}_kernel_trace_256cubed_summary.txt | head -12; tail -6 $out/${tag}_c4_inloop_probe.txt
