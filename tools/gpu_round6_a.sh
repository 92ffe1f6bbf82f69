#!/bin/bash
# round 6, first GPU pass: smoke, the tests this round touched, the L2 counter passes, a bench line without the secondary workloads
export PYTHONPATH=$PWD
out=$PWD/gpurun_out
mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke()" > $out/r06a_smoke.log 2>&1; echo "smoke rc=$?"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "lossless or do_not_ascend or chip_solve_arguments or somebody_else_holds or drop_in or test_chip_solve_equals" > $out/r06a_tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/r06a_tests.log
timeout 1500 bash tools/pmc_chip_l2.sh r06 > $out/r06a_pmc_l2.log 2>&1; echo "pmc rc=$?"
cp profiles/r06_chip_l2_counters.md profiles/r06_chip_kernel_stats.csv profiles/pmc_traffic.json $out/ 2>/dev/null
timeout 900 python bench.py --no-extra --no-cpu-baseline > $out/r06a_bench.json 2> $out/r06a_bench.err; echo "bench rc=$?"
tail -5 $out/r06a_smoke.log; tail -30 $out/r06a_pmc_l2.log; python - <<'P'
import json
d=json.load(open('gpurun_out/r06a_bench.json')); r=d['roofline']
print(d['value'], d['ms_per_step'])
print({k:r[k] for k in r if k not in ('measured_stream_gbs','hbm_bound_256cubed','regime','kernel','streaming_spmv_kernel')})
P
