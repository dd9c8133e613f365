# randomised parity beyond the suite's seeds, on the default path selection and with either exact-tick pipeline forced
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fuzz2; mkdir -p $O; cd $R
timeout -k 10 500 python scripts/fuzz_campaign.py 1500 400 > $O/default.txt 2>&1; echo "default rc=$?"; tail -1 $O/default.txt
DMX_SMALL_EXACT=2 timeout -k 10 400 python scripts/fuzz_campaign.py 1000 200 > $O/small_always.txt 2>&1; echo "small-always rc=$?"; tail -1 $O/small_always.txt
DMX_SMALL_EXACT=0 timeout -k 10 400 python scripts/fuzz_campaign.py 600 100 > $O/stage_per_launch.txt 2>&1; echo "stage-per-launch rc=$?"; tail -1 $O/stage_per_launch.txt
grep -h "FAIL" $O/*.txt | head
