#!/bin/bash
# L2-level and issue counters of the whole-chip solve kernel + the kernel trace of its three variants:
#   gpurun -- 'bash tools/pmc_chip_l2.sh r06'
# One rocprofv3 pass per counter group (--pmc with --kernel-trace only; the program directly after `--`); a group the device does not
# know is reported and skipped.
tag=${1:-r06}
export PYTHONPATH=$PWD
repo=$PWD
out=$PWD/gpurun_out/${tag}_pmc_chip_l2
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $out/available_counters.txt 2>&1
python3 $repo/tools/pmc_chip_l2_run.py > $out/run.log 2>$out/run.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $repo/tools/pmc_chip_l2_run.py --variants > $out/trace.log 2>&1
echo "trace rc=$?" >> $out/passes.txt
i=0
while read -r c; do
    [ -z "$c" ] && continue
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pass$i -- python3 $repo/tools/pmc_chip_l2_run.py > $out/pass$i.log 2>&1
    echo "pass $i [$c] rc=$?" >> $out/passes.txt
done <<'EOF'
TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
TCC_READ_sum TCC_WRITE_sum
TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES
SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES
SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE
GRBM_GUI_ACTIVE GRBM_COUNT
FETCH_SIZE
WRITE_SIZE
EOF
find $out -type f ! -name '*counter_collection.csv' ! -name '*kernel_stats.csv' ! -name '*.log' ! -name '*.txt' ! -name '*.err' -delete
cd $repo && python3 tools/pmc_chip_l2_report.py $out $tag > $out/summary.md 2>&1
cat $out/passes.txt; cat $out/summary.md
