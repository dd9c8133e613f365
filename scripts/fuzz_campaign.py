"""One-off campaign: the randomised parity tests of tests/test_gpu_fuzz.py over seeds beyond the suite's."""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib.util
spec = importlib.util.spec_from_file_location("fz", os.path.join(ROOT, "tests", "test_gpu_fuzz.py")); fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
import torch.distributed as dist, socket
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
dist.init_process_group("nccl", rank=0, world_size=1)
n_scene, n_seq = int(sys.argv[1]), int(sys.argv[2])
bad = []
t0 = time.time()
for seed in range(64, 64 + n_scene):
    try:
        fz.test_random_scene_matches_oracle(seed)
    except Exception as e:      # noqa: BLE001
        bad.append(("scene", seed, str(e)[:300])); print("FAIL scene", seed, str(e)[:300], flush=True)
    if seed % 100 == 0: print("scene seed", seed, f"{time.time()-t0:.0f}s", flush=True)
for seed in range(32, 32 + n_seq):
    try:
        fz.test_random_call_sequences_match_oracle.__wrapped__(seed, dist) if hasattr(fz.test_random_call_sequences_match_oracle, "__wrapped__") else fz.test_random_call_sequences_match_oracle(seed, dist)
    except Exception as e:      # noqa: BLE001
        bad.append(("sequence", seed, str(e)[:300])); print("FAIL sequence", seed, str(e)[:300], flush=True)
    if seed % 50 == 0: print("sequence seed", seed, f"{time.time()-t0:.0f}s", flush=True)
print(f"campaign: {n_scene} scene seeds + {n_seq} sequence seeds, {len(bad)} failures, {time.time()-t0:.0f} s")
for b in bad: print(b)
dist.destroy_process_group()
sys.exit(1 if bad else 0)
