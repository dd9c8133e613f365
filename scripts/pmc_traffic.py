"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into profiles/hbm_pmc_<kind>_<dtype>_<n>.json."""
import csv, glob, json, statistics, sys
kind, dtype, n, kernel, fetch_dir, write_dir, out = sys.argv[1:8]
def med(d, name):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if kernel in r["Kernel_Name"] and r["Counter_Name"] == name]
    return {"dispatches": len(v), "median_KB": statistics.median(v), "min_KB": min(v), "max_KB": max(v)}
o = {"kernel": kernel, "FETCH_SIZE": med(fetch_dir, "FETCH_SIZE"), "WRITE_SIZE": med(write_dir, "WRITE_SIZE"),
     "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes over bench.py. gfx950: FETCH_SIZE counts half the "
             "bytes of 16 B/lane coalesced streams (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact. traffic = (2*FETCH+WRITE)*1024 B per launch."}
o["traffic_bytes_per_launch"] = (2 * o["FETCH_SIZE"]["median_KB"] + o["WRITE_SIZE"]["median_KB"]) * 1024
json.dump(o, open(out, "w"), indent=1)
print(out, o["traffic_bytes_per_launch"] / 1e6, "MB")
