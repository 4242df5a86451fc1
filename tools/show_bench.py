"""Print the essentials of a bench.py JSON line (file argument)."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline", round(d["value"], 2), d["unit"], round(d["ms_per_step"], 3), "ms/step")
cb = d.get("cpu_baseline") or {}
if cb:
    print("cpu", cb.get("value"), cb.get("sample"))
    for l in cb.get("legs", []):
        print("  leg", l.get("threads"), l.get("value"), l.get("sample"), l.get("passes_s"), l.get("error"))
for s in d.get("secondary") or []:
    print("  sec", s["args"], s.get("value"), s.get("ms_per_step"), s.get("error"))
r = d.get("roofline")
if r:
    print({k: r.get(k) for k in ("bound", "frac", "kernel", "avg_us", "avg_us_net", "hbm_frac", "hbm_frac_net", "mfma_frac", "event_bracket_us")})
    for f in r["families"]:
        print("   ", f["family"], f["ms_per_step"], f["launches_per_step"], f["hbm_frac"], f["mfma_frac"])
print("conv_bytes_per_step", d.get("conv_bytes_per_step"), "ideal", d.get("ideal_fused_bytes_per_step"))
