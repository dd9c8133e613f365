# round 4: randomised parity beyond the suite's seeds on the final build -- the general campaign of round 3 (default path selection, either
# exact-tick pipeline forced, speculation off) and the hull-against-map campaign (scripts/fuzz_hulls_r04.py), the latter also through the
# wavefront-per-body kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fuzz4; mkdir -p $O; cd $R
timeout -k 10 420 python scripts/fuzz_campaign.py 1500 400 > $O/default.txt 2>&1; echo "default rc=$?"; tail -1 $O/default.txt
DMX_SMALL_EXACT=2 timeout -k 10 300 python scripts/fuzz_campaign.py 800 150 > $O/small_always.txt 2>&1; echo "small-always rc=$?"; tail -1 $O/small_always.txt
DMX_SMALL_EXACT=0 timeout -k 10 300 python scripts/fuzz_campaign.py 600 100 > $O/stage_per_launch.txt 2>&1; echo "stage-per-launch rc=$?"; tail -1 $O/stage_per_launch.txt
DMX_HYBRID_PAIRS=0 DMX_REGS_BY_CONTACT=0 timeout -k 10 300 python scripts/fuzz_campaign.py 600 100 > $O/r03_forms.txt 2>&1; echo "no hybrid pipeline, sweeps row by row rc=$?"; tail -1 $O/r03_forms.txt
timeout -k 10 600 python scripts/fuzz_hulls_r04.py 3000 1000 > $O/hulls.txt 2>&1; echo "hulls rc=$?"; tail -1 $O/hulls.txt
DMX_HULL_WAVE_PER_BODY=1 timeout -k 10 300 python scripts/fuzz_hulls_r04.py 600 5000 > $O/hulls_wave.txt 2>&1; echo "hulls, a wavefront per body rc=$?"; tail -1 $O/hulls_wave.txt
timeout -k 10 400 python scripts/fuzz_far_r04.py 3000 1000 > $O/far.txt 2>&1; echo "far from the origin, boxes rc=$?"; tail -1 $O/far.txt
grep -h "FAIL" $O/*.txt | head
