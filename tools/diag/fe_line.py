import json, sys
d = json.loads(sys.stdin.read())
print(sys.argv[1] if len(sys.argv) > 1 else "", d["dtype"], "ms/step", d["ms_per_step"], "value", d["value"], d["unit"],
      "| layers route ms", d["config"].get("layer_by_layer_exact_fp32_ms_per_step"))
