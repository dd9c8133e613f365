"""one-screen summary of a bench.py JSON line: python scripts/bench_summary.py <file>"""
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print(f"headline {d['value'] / 1e9:.1f} G {d['unit']}, {d['ms_per_step'] * 1e3:.1f} us/tick, frac {r['frac']:.3f} ({r['bound']})")
for k, v in d.get("configs", {}).items():
    if "error" in v:
        print(k, "ERROR", v["error"]); continue
    rr = v["roofline"]
    print(f"{k}: {v['ms_per_step'] * 1e3:.1f} us/tick (mean {v['ms_per_step_mean'] * 1e3:.1f}), {rr['bound']} frac {rr.get('frac')}")
    if "hull_pairs_on" in v: print("   hull_pairs_on:", f"{v['hull_pairs_on']['ms_per_step'] * 1e3:.1f} us/tick")
for k in ("f64", "hbm_resident", "cpu_baseline"):
    if k in d: print(k, {a: b for a, b in d[k].items() if a in ("value", "ms_per_step", "unit", "cores", "kind")})
if "reference_pen" in d:
    for k, v in d["reference_pen"].items():
        print("reference pen", k, v.get("error") or f"{v['ms_per_step'] * 1e3:.0f} us/tick, contacts {v['contacts_last_tick']}")
