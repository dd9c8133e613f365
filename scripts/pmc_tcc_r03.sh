#!/bin/bash
# Where does the HBM-resident free-flight pass lose against a float4 copy of the same bytes?  L2 (TCC) memory-side counters of
# integrate_free at 16 Mi f32 bodies (bench.py --side 4096) and of the copy / in-place / tile-pattern kernels of
# scripts/ubench_inplace.hip at the same size: request counts, DRAM credit stalls, write-request stalls, request levels
# (occupancy: level / requests = cycles in flight) -- one rocprofv3 --pmc pass per group of four (the TCC has four slots).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_tcc; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
SETS=("TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum"
        "TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"
        "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum TCC_REQ_sum"
        "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_HIT_sum TCC_MISS_sum"
        "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_64B_sum")
g=0
for grp in "${SETS[@]}"; do
  rocprofv3 --pmc $grp --output-format csv -d $O/bench_g$g -- python3 $R/bench.py --side 4096 --no-extras --no-cpu-baseline --no-body-collisions --steps 30 --warmup 5 > $O/bench_g$g.log 2>&1 || { echo "bench pass $g failed"; tail -5 $O/bench_g$g.log; exit 1; }
  rocprofv3 --pmc $grp --output-format csv -d $O/ubench_g$g -- $R/scripts/_build/ubench_inplace 4096 > $O/ubench_g$g.log 2>&1 || { echo "ubench pass $g failed"; tail -5 $O/ubench_g$g.log; exit 1; }
  g=$((g+1))
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- python3 $R/bench.py --side 4096 --no-extras --no-cpu-baseline --no-body-collisions --steps 30 --warmup 5 > $O/bench_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ubench_stats -- $R/scripts/_build/ubench_inplace 4096 > $O/ubench_stats.log 2>&1
python3 $R/scripts/pmc_tcc_table.py $O > $R/gpurun_out/r03_tcc_counters_16Mi.txt; cat $R/gpurun_out/r03_tcc_counters_16Mi.txt
