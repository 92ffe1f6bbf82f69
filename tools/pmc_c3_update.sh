#!/bin/bash
# Memory-side traffic of the caller-order IC(0) update on the config-3 stand-in (records form, then the CSR-stream form of the
# sync-free solves):  gpurun -- 'bash tools/pmc_c3_update.sh r04'   ->  profiles/<tag>_c3_update_counters.md
tag=${1:-r04}
export PYTHONPATH=$PWD
repo=$PWD
out=$PWD/gpurun_out/${tag}_pmc_c3
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for form in records stream; do
    if [ $form = stream ]; then export DPCG_SF_STREAM=1; else export DPCG_SF_STREAM=0; fi
    i=0
    for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
             "TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
        i=$((i+1))
        timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$form/pass$i -- python3 $repo/tools/trace_run_c3.py > $out/$form.pass$i.log 2>&1
        echo "$form pass $i rc=$?" >> $out/passes.txt
    done
done
find $out -type f ! -name '*counter_collection.csv' ! -name '*kernel_trace.csv' ! -name '*.log' ! -name '*.txt' -delete
cd $repo && python3 tools/pmc_c3_update_report.py $out $tag > $out/summary.md 2>&1
cat $out/passes.txt; cat $out/summary.md
