#!/bin/bash
# Kernel trace + PMC traffic of the whole-chip solve:  gpurun -- 'bash tools/pmc_chip.sh r05'
tag=${1:-r05}
export PYTHONPATH=$PWD
repo=$PWD
out=$PWD/gpurun_out/${tag}_pmc_chip
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $repo/tools/pmc_chip_run.py > $out/run.log 2>$out/run.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $repo/tools/pmc_chip_run.py > $out/trace.log 2>&1
echo "trace rc=$?" >> $out/passes.txt
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pass$i -- python3 $repo/tools/pmc_chip_run.py > $out/pass$i.log 2>&1
    echo "pass $i rc=$?" >> $out/passes.txt
done
find $out -type f ! -name '*counter_collection.csv' ! -name '*kernel_stats.csv' ! -name '*.log' ! -name '*.txt' ! -name '*.err' -delete
cd $repo && python3 tools/pmc_chip_report.py $out $tag > $out/summary.md 2>&1
cat $out/passes.txt; cat $out/summary.md
