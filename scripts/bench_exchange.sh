# one-rank rehearsal of the N>1 loop (the all-gather degenerates to a copy): cost of the exchange and of carrying the collision proof
cd $GRAFT_REPO_ROOT
p='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.2f us/tick  %.1f Gbs/s  | %s | %s" % (d["ms_per_step"]*1e3, d["value"]/1e9, d["config"]["parallelism"][60:200], d["config"]["collide"][:90]))'
echo -n "no exchange (C loop)            : "; python bench.py --no-cpu-baseline 2>/dev/null | python -c "$p"
echo -n "exchange at chunk ends, collide : "; python bench.py --no-cpu-baseline --force-exchange 2>/dev/null | python -c "$p"
echo -n "every tick, graph 16, collide   : "; python bench.py --no-cpu-baseline --force-exchange --exchange-every-tick 2>/dev/null | python -c "$p"
echo -n "every tick, eager, collide      : "; python bench.py --no-cpu-baseline --force-exchange --exchange-every-tick --graph-steps 0 2>/dev/null | python -c "$p"
echo -n "every tick, graph 16, no collide: "; python bench.py --no-cpu-baseline --force-exchange --no-body-collisions 2>/dev/null | python -c "$p"
echo -n "every tick, eager, no collide   : "; python bench.py --no-cpu-baseline --force-exchange --graph-steps 0 --no-body-collisions 2>/dev/null | python -c "$p"
