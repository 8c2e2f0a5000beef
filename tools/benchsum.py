import json, sys
d = json.load(open(sys.argv[1]))
print("it/s %.2f  ms/step %.2f  path %s  frac_peak %.3f" % (d["value"], d["ms_per_step"], d["config"]["kernel_path"], d["iteration"]["frac_f32_peak"]))
for k, v in d["kernels"].items():
    r = d.get("roofline_by_kernel", {}).get(k)
    extra = ("%-4s %s %.3g %s (frac %.2f)" % (v.get("family"), r["bound"], r["achieved"], r["unit"], r["frac"])) if r else ""
    print("  %-12s avg %.3f ms  %s" % (k, v["avg_ms"], extra))
for lab in ("direct_variant", "fft_variant"):
    if lab in d and "value" in d[lab]:
        v = d[lab]
        print("  %-15s %.2f it/s  %.2f ms/step  x%.2f of main  dW %.1e" % (lab, v["value"], v["ms_per_step"], v["speed_relative_to_main"], v["W_max_rel_diff_vs_main"]))
