# rocprofv3 evidence for the round: kernel-trace stats for configs[1] and configs[2], PMC HBM traffic for configs[1]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2 -- python3 $R/bench.py --steps 500 --warmup 50 --no-cpu-baseline > $O/c2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -- python3 $R/bench.py --config 3 --steps 200 --warmup 20 --no-cpu-baseline > $O/c3.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c2_fetch -- python3 $R/bench.py --steps 40 --warmup 8 --no-cpu-baseline > $O/c2_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c2_write -- python3 $R/bench.py --steps 40 --warmup 8 --no-cpu-baseline > $O/c2_write.log 2>&1
cd $R
python3 scripts/pmc_traffic.py free f32 1048576 integrate_free $O/c2_fetch $O/c2_write $O/hbm_pmc_free_f32_1048576.json
cat $O/c2/*/*kernel_stats.csv | cut -c1-200
cat $O/c3/*/*kernel_stats.csv | cut -c1-200
tail -1 $O/c2.log | cut -c1-400
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -- python3 $R/bench.py --config 5 --steps 200 --warmup 20 --no-cpu-baseline > $O/c5.log 2>&1
cd $R
cat "$(ls -t $O/c5/*/*kernel_stats.csv | head -1)" | cut -c1-200
tail -1 $O/c5.log | cut -c1-600
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c3_fetch -- python3 $R/bench.py --config 3 --steps 40 --warmup 8 --no-cpu-baseline > $O/c3_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c3_write -- python3 $R/bench.py --config 3 --steps 40 --warmup 8 --no-cpu-baseline > $O/c3_write.log 2>&1
cd $R
python3 scripts/pmc_traffic.py plane f32 262144 step_plane $O/c3_fetch $O/c3_write $O/hbm_pmc_plane_f32_262144.json
