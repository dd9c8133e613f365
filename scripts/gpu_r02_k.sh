# round-2 evidence for the exact tick on small scenes: configs[0] through the batch path, stage times, piles, the ODE API face
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02k; mkdir -p $O; cd $R
python scripts/time_config1.py 2>&1 | grep -v amdgpu.ids | tee $O/r02_config1_batch_path.txt
DMX_EXS_TIMING=1 python scripts/time_config1.py 2>&1 | grep -v amdgpu.ids > $O/r02_small_scene_stages.txt; tail -18 $O/r02_small_scene_stages.txt
DMX_SMALL_EXACT=0 python scripts/time_config1.py 2>&1 | grep -v amdgpu.ids | grep config1 | sed 's/^/DMX_SMALL_EXACT=0 (a stage per launch): /' | tee -a $O/r02_config1_batch_path.txt
python scripts/time_piles.py 2>&1 | grep -v amdgpu.ids | tee $O/r02_exact_tick_piles.txt
bash scripts/profile_config1.sh > $O/prof_config1.log 2>&1; cp $R/gpurun_out/prof_config1/busy.txt $O/r02_config1_trace.txt; cat $O/r02_config1_trace.txt | head -12
bash scripts/time_compat_exact.sh > $O/r02_compat_both_steppers.txt 2>&1; grep "reference scene" $O/r02_compat_both_steppers.txt
