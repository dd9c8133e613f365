# round 3: randomised parity beyond the suite's seeds on the default path selection (speculative small exact ticks, LDS grid, workgroup
# level schedules, rows in registers), with the speculation off, and with either exact-tick pipeline forced
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fuzz3; mkdir -p $O; cd $R
timeout -k 10 420 python scripts/fuzz_campaign.py 1500 400 > $O/default.txt 2>&1; echo "default rc=$?"; tail -1 $O/default.txt
DMX_SMALL_EXACT=2 timeout -k 10 300 python scripts/fuzz_campaign.py 1000 200 > $O/small_always.txt 2>&1; echo "small-always rc=$?"; tail -1 $O/small_always.txt
DMX_SMALL_EXACT=0 timeout -k 10 300 python scripts/fuzz_campaign.py 600 100 > $O/stage_per_launch.txt 2>&1; echo "stage-per-launch rc=$?"; tail -1 $O/stage_per_launch.txt
DMX_SPECULATE=0 DMX_SMALL_LDS_GRID=0 timeout -k 10 200 python scripts/fuzz_campaign.py 600 100 > $O/no_speculation.txt 2>&1; echo "no speculation, bucket grid rc=$?"; tail -1 $O/no_speculation.txt
grep -h "FAIL" $O/*.txt | head
