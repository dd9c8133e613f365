# same-box A/B of several builds of the library on configs[4]: gpurun_ab/libode_mi355_<tag>.so against the tree's own ("new").
# The box's copy of the tree is scratch: the library file is swapped in place between the runs, round-robin, two rounds.
R=$GRAFT_REPO_ROOT; cd /tmp
cp $R/rl-ode-physics_amd/libode_mi355.so /tmp/lib_new.so
run() { python3 $R/bench.py --config 5 --no-extras --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | python3 -c "import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 ms_per_step %.5f' % o['ms_per_step'])"; }
for rep in 1 2; do
  cp /tmp/lib_new.so $R/rl-ode-physics_amd/libode_mi355.so; run new
  for f in $R/gpurun_ab/libode_mi355_*.so; do t=$(basename $f .so); cp $f $R/rl-ode-physics_amd/libode_mi355.so; run ${t#libode_mi355_}; done
done
cp /tmp/lib_new.so $R/rl-ode-physics_amd/libode_mi355.so
