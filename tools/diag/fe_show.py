import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ("metric","value","ms_per_step","dtype")}); print("layers route ms:", d["config"]["layer_by_layer_exact_fp32_ms_per_step"]); print(d.get("roofline"))
for r in d["kernels"][:14]: print(r["kernel"], r["calls_per_step"], r["ms_per_step"], r["TFLOPps"], r["GBps"], r["frac"])
