"""Fold the passes of scripts/pmc_tcc_r03.sh into one table: per kernel, its average duration and every counter per launch."""
import csv, glob, statistics, sys
O = sys.argv[1]
want = {"integrate_free": "integrate_free (product, in place)", "k_copy": "float4 copy, out of place", "k_scale_inplace": "float4 read-modify-write in place",
        "k_tile<30>": "tile pattern 17r/13w (first of: in place, out of place)", "k_split": "split dense layout (first of: out of place, in place)"}
dur = {}
for tag in ("bench", "ubench"):
    for f in glob.glob(f"{O}/{tag}_stats/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for k in want:
                if k in r["Name"] and k not in dur:
                    dur[k] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
vals = {}
for f in glob.glob(f"{O}/*_g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k in want:
            if k in r["Kernel_Name"]:
                vals.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
counters = sorted({c for (_, c) in vals})
print("16 777 216 f32 bodies (2 013 MB algorithmic per pass).  Per launch, median over the launches of each pass; rocprofv3 --pmc, gfx950.")
print(f"{'counter':42s}" + "".join(f"{k:>22s}" for k in want))
print(f"{'average duration, us (kernel trace)':42s}" + "".join(f"{dur.get(k, (float('nan'), 0))[0]:22.1f}" for k in want))
for c in counters:
    print(f"{c:42s}" + "".join(f"{statistics.median(vals[(k, c)]) if (k, c) in vals else float('nan'):22.4g}" for k in want))
for k, v in want.items():
    print(f"  {k:18s} = {v}")
