#!/bin/bash
# PMC traffic of the SpMV on the config-3 mesh systems:  gpurun -- 'bash tools/pmc_mesh.sh r04'
tag=${1:-r04}
export PYTHONPATH=$PWD
repo=$PWD
out=$PWD/gpurun_out/${tag}_pmc_mesh
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $repo/tools/pmc_mesh_run.py > $out/run.log 2>$out/run.err
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"; do
    i=$((i+1))
    timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pass$i -- python3 $repo/tools/pmc_mesh_run.py > $out/pass$i.log 2>&1
    echo "pass $i rc=$?" >> $out/passes.txt
done
find $out -type f ! -name '*counter_collection.csv' ! -name '*.log' ! -name '*.txt' ! -name '*.err' -delete
cd $repo && python3 tools/pmc_mesh_report.py $out/pass1 $out/pass2 $out/pass3 $out/run.log $tag > $out/summary.md 2>&1
cat $out/passes.txt; cat $out/summary.md
