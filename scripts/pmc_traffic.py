"""Fold rocprofv3 passes over bench.py into profiles/hbm_pmc_<kind>_<dtype>_<n>.json: two --pmc passes (FETCH_SIZE, WRITE_SIZE)
and one --kernel-trace --stats pass (the kernel's average duration), stamped with the git blob hash of csrc/dmx_kernels.hip
they were measured on -- bench.py refuses the numbers when the kernel source has changed since."""
import csv, glob, hashlib, json, os, statistics, sys
kind, dtype, n, kernel, fetch_dir, write_dir, stats_dir, out = sys.argv[1:9]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def med(d, name):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if kernel in r["Kernel_Name"] and r["Counter_Name"] == name]
    return {"dispatches": len(v), "median_KB": statistics.median(v), "min_KB": min(v), "max_KB": max(v)}
def kernel_us(d):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if kernel in r["Name"]]
    rows.sort(key=lambda r: -int(r["Calls"]))
    return float(rows[0]["AverageNs"]) / 1e3, int(rows[0]["Calls"]), rows[0]["Name"][:120]
data = open(os.path.join(ROOT, "rl-ode-physics_amd", "csrc", "dmx_kernels.hip"), "rb").read()
blob = hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()
us, calls, name = kernel_us(stats_dir)
o = {"kernel": kernel, "kernel_instance": name, "FETCH_SIZE": med(fetch_dir, "FETCH_SIZE"), "WRITE_SIZE": med(write_dir, "WRITE_SIZE"),
     "rocprof_kernel_us": us, "rocprof_kernel_calls": calls, "kernels_blob": blob,
     "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes over bench.py, and a --kernel-trace --stats pass. gfx950: "
             "FETCH_SIZE counts half the bytes of 16 B/lane coalesced streams (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact. "
             "traffic = (2*FETCH+WRITE)*1024 B per launch. kernels_blob = git hash-object of csrc/dmx_kernels.hip at measurement time."}
o["traffic_bytes_per_launch"] = (2 * o["FETCH_SIZE"]["median_KB"] + o["WRITE_SIZE"]["median_KB"]) * 1024
json.dump(o, open(out, "w"), indent=1)
print(out, o["traffic_bytes_per_launch"] / 1e6, "MB", us, "us")
