# rocprofv3 evidence for round 4: kernel-trace stats + PMC HBM traffic for configs[1] (f32, f64, 16 Mi f32) and configs[2];
# kernel-trace stats for configs[4] and the static-floor scene; the configs[0] kernel trace.  Summaries are copied into
# profiles/ by hand afterwards (gpurun_out/ is scratch).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r04; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-extras"
prof() {   # name  kernel  kind dtype n  -- bench flags
    name=$1; kernel=$2; kind=$3; dtype=$4; n=$5; shift 5
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/${name}_stats -- $B "$@" --steps 300 --warmup 40 > $O/${name}_stats.log 2>&1 || return 1
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${name}_fetch -- $B "$@" --steps 40 --warmup 8 > $O/${name}_fetch.log 2>&1 || return 1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${name}_write -- $B "$@" --steps 40 --warmup 8 > $O/${name}_write.log 2>&1 || return 1
    (cd $R && python3 scripts/pmc_traffic.py $kind $dtype $n $kernel $O/${name}_fetch $O/${name}_write $O/${name}_stats $O/hbm_pmc_${kind}_${dtype}_${n}.json)
    cp "$(ls $O/${name}_stats/*/*kernel_stats.csv | head -1)" $O/r04_${name}_kernel_stats.csv
    tail -1 $O/${name}_stats.log | cut -c1-200
}
prof c2_f32 integrate_free free f32 1048576 || exit 1
prof c2_f64 integrate_free free f64 1048576 --dtype f64 || exit 1
prof c2_f32_16Mi integrate_free free f32 16777216 --side 4096 || exit 1
prof c3_f32 step_plane plane f32 262144 --config 3 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5_stats -- $B --config 5 --steps 200 --warmup 20 > $O/c5_stats.log 2>&1 || exit 1
cp "$(ls $O/c5_stats/*/*kernel_stats.csv | head -1)" $O/r04_c5_f32_kernel_stats.csv; tail -1 $O/c5_stats.log | cut -c1-200
rocprofv3 --kernel-trace --stats --output-format csv -d $O/floor_stats -- python3 $R/scripts/time_floor.py floor > $O/floor_stats.log 2>&1 || exit 1
cp "$(ls $O/floor_stats/*/*kernel_stats.csv | head -1)" $O/r04_static_floor_kernel_stats.csv; grep "ms/tick" $O/floor_stats.log
python3 $R/scripts/time_floor.py > $O/r04_static_floor_and_hulls.txt 2>&1; cat $O/r04_static_floor_and_hulls.txt
python3 $R/scripts/time_config1.py > $O/r04_config1_batch_path.txt 2>&1; cat $O/r04_config1_batch_path.txt
ls $O/*.json $O/*.csv
