# functional rehearsal of bench.py's N>1 path on a one-GPU box: all ranks on cuda:0, collectives staged through gloo
cd $GRAFT_REPO_ROOT
run() { n=$1; shift; timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus $n --rehearse-on-one-gpu --no-cpu-baseline "$@" 2>gpurun_out/rehearse_err.txt | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["n_gpus"], "ranks: %.2f us/tick, %.1f G body-steps/s |" % (d["ms_per_step"]*1e3, d["value"]/1e9), d["config"]["parallelism"][45:], "|", d["config"]["collide"][:100])' || { tail -20 gpurun_out/rehearse_err.txt; return 1; }; }
run 2 --side 512 --steps 300 --warmup 30 &&
run 4 --side 256 --steps 300 --warmup 30 &&
run 2 --side 512 --steps 200 --warmup 20 --exchange-every-tick --graph-steps 0 &&
run 2 --config 3 --side 256 --steps 200 --warmup 20 --graph-steps 0
