#!/bin/bash
# Regenerates the per-round evidence on the GPU box: run as `gpurun --timeout 2400 -- 'bash tools/profile_round.sh r01'`.
# Outputs land under gpurun_out/<tag>_*; tools/pmc_report.py and tools/trace_gaps.py turn them into profiles/.
tag=${1:-r01}
export PYTHONPATH=$PWD
out=$PWD/gpurun_out
mkdir -p $out
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf $out/${tag}_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python3 $OLDPWD/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $out/${tag}_stats.log 2>&1
rm -rf $out/${tag}_stats_c4
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats_c4 -- python3 $OLDPWD/tools/trace_run_c4.py > $out/${tag}_stats_c4.log 2>&1
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"; do
    name=${tag}_pmc_$(echo $c | cut -c1-12 | tr ' ' '_')
    rm -rf $out/$name
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$name -- python3 $OLDPWD/tools/pmc_run.py > $out/$name.log 2>&1
done
rm -rf $out/${tag}_c3_trace
rocprofv3 --kernel-trace --output-format csv -d $out/${tag}_c3_trace -- python3 $OLDPWD/tools/trace_run_c3.py > $out/${tag}_c3_trace.log 2>&1
rm -rf $out/${tag}_c3mc_trace
rocprofv3 --kernel-trace --output-format csv -d $out/${tag}_c3mc_trace -- python3 $OLDPWD/tools/trace_run_c3.py multicolor > $out/${tag}_c3mc_trace.log 2>&1
rm -rf $out/${tag}_cnn_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_cnn_stats -- python3 $OLDPWD/tools/trace_run_cnn.py > $out/${tag}_cnn_stats.log 2>&1
# keep the merged-back volume small: only the CSVs that the reports read
find $out/${tag}_stats $out/${tag}_stats_c4 $out/${tag}_cnn_stats $out/${tag}_c3_trace $out/${tag}_c3mc_trace $out/${tag}_pmc_* -type f ! -name '*kernel_stats.csv' ! -name '*counter_collection.csv' ! -name '*kernel_trace.csv' -delete
ls -la $out/${tag}_stats/* | head
