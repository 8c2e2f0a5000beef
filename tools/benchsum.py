import json,sys
d=json.load(open(sys.argv[1]))
print("it/s %.2f  ms/step %.2f  path %s  frac_peak %.3f" % (d["value"], d["ms_per_step"], d["config"]["kernel_path"], d["iteration"]["frac_f32_peak"]))
for k,v in d["kernels"].items(): print("  %-12s avg %.3f ms  %s" % (k, v["avg_ms"], ("%.1f TF" % v["tflops"]) if "tflops" in v else ""))
