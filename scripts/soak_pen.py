"""One-off soak: the reference's pen for many ticks, bit for bit against the oracle (tests/test_gpu_large_island.py's check, longer).
usage: soak_pen.py bodies ticks dtype [pipeline]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import tests.test_gpu_large_island as t
n, ticks, dtype = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
pipeline = int(sys.argv[4]) if len(sys.argv) > 4 else None
most, st = t._run_both(n, dtype, ticks, pipeline=pipeline)
print(f"pen {n} bodies, {ticks} ticks, {dtype}, pipeline {pipeline}: identical to the oracle; most contacts {most}; {st}", flush=True)
