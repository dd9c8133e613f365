# same-box A/B at an HBM-resident size (16 Mi bodies, 2 GB of state): the contact-free pass in place vs alternating slabs (DMX_OOP=1)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for O in 0 1; do
  DMX_OOP=$O python3 bench.py --side 4096 --steps 200 --warmup 20 --no-extras --no-cpu-baseline --no-body-collisions 2>/dev/null | python3 -c "
import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DMX_OOP=$O rep $rep ms_per_step', round(o['ms_per_step'],5), 'frac', round(o['roofline']['frac'],4), 'mean', round(o['timing']['ms_per_step_mean'],5))"
done; done
