#!/bin/bash
export PYTHONPATH=$PWD
repo=$PWD
out=$PWD/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in graphalways; do
  rm -rf $out/r06_c4_$v
  export DPCG_GRAPH_ALWAYS=1
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/r06_c4_$v -- python3 $repo/tools/trace_run_c4.py > $out/r06_c4_$v.log 2>&1
  tr=$(ls $out/r06_c4_$v/*/*kernel_trace.csv | head -1)
  python3 $repo/tools/trace_gaps.py $tr > $out/r06_kernel_trace_256cubed_summary_$v.txt 2>&1
  find $out/r06_c4_$v -type f -delete
  echo "== $v"; head -6 $out/r06_kernel_trace_256cubed_summary_$v.txt; grep "^gap" $out/r06_kernel_trace_256cubed_summary_$v.txt | head -8; tail -2 $out/r06_c4_$v.log
done
