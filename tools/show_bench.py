"""One line per bench.py JSON line: python tools/show_bench.py a.json b.json ..."""
import json
import sys

for f in sys.argv[1:]:
    d = json.load(open(f))
    t = d["device_ms_timed_region"]
    r = d["roofline"]
    p = d.get("parity") or {}
    print(f, f"{d['value'] / 1e6:.2f} M sites/s", f"{d['ms_per_step']:.1f} ms/step x {d['steps']}", f"frac {r['frac']:.3f} achieved {r['achieved']:.0f}",
          {k: round(v) for k, v in t.items() if k.endswith("_ms") and v > 0}, "const tiles", r.get("tiles_constant"), "listed", r.get("tiles_conv4_on_listed_rows"),
          "tiles", r.get("tiles"), "parity", p.get("max_abs_dp_vs_oracle"), p.get("sites_checked"), "cpu", (d.get("cpu_baseline") or {}).get("value"),
          "e2e", (d.get("end_to_end") or {}).get("value"))
