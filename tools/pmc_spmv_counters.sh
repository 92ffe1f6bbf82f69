#!/bin/bash
# What bounds k_spmv_tile on the HBM-bound 256^3 system?  rocprofv3 --pmc passes (SQ / TA-TCP-TD / TCC blocks, each in its
# own run, --kernel-trace only beside --pmc) on tools/trace_run_c4.py; k_update_r and k_update_xp_deferred of the same
# solve are the in-run comparison (streaming kernels that reach ~6 TB/s).  Summary: tools/pmc_counters_report.py.
#   gpurun -- 'bash tools/pmc_spmv_counters.sh r03'
tag=${1:-r03}
workload=${2:-tools/trace_run_c4.py}
export PYTHONPATH=$PWD
repo=$PWD
out=$PWD/gpurun_out/${tag}_pmc_spmv
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
while read -r counters; do
    [ -z "$counters" ] && continue
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $out/pass$i -- python3 $repo/$workload > $out/pass$i.log 2>&1
    echo "pass $i rc=$? : $counters" >> $out/passes.txt
done <<'LIST'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_BUSY_CU_CYCLES
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum GRBM_GUI_ACTIVE
TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum
TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TD_STORE_WAVEFRONT_sum
TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
TCC_BUSY_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum
MeanOccupancyPerActiveCU
MemUnitBusy MemUnitStalled
LIST
find $out -type f ! -name '*counter_collection.csv' ! -name '*kernel_trace.csv' ! -name '*.log' ! -name 'passes.txt' -delete
cd $repo && python3 tools/pmc_counters_report.py $out > $out/summary.md 2>&1
cat $out/passes.txt
tail -60 $out/summary.md
