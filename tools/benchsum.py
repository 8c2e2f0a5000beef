"""Prints the headline numbers of one bench.py JSON line (file given as argv[1])."""
import json
import sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("it/s %.2f  ms/step %.3f  path %s" % (d["value"], d["ms_per_step"], d["config"]["kernel_path"]))
for k, v in d["kernels"].items():
    print("  %-12s %-6s avg %.3f ms x %d" % (k, v.get("family"), v["avg_ms"], v["launches"]))
for k in ("exact_f32_variant", "direct_variant", "fft_variant"):
    if k in d and "value" in d[k]:
        print("  %s: %.2f it/s  %s" % (k, d[k]["value"], {a: round(b, 3) for a, b in d[k]["kernels_ms"].items()}))
if "parity" in d:
    p = d["parity"]
    print("  parity (%d samples): dW %.2e dH %.2e dE %.2e" % (p["samples"], p["W_rel_diff_vs_oracle"], p["H_rel_diff_vs_oracle"], p["energy_gap_vs_oracle"]))
r = d["roofline"]
print("  roofline: %s %s %.1f / %.0f %s = %.3f" % (r["kernel"], r["family"], r["achieved"], r["peak"], r["unit"], r["frac"]))
