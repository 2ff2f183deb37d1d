"""Compare two PN2_TIMER_DETAIL bench JSONs: python tools/cmp_detail.py new.json old.json"""
import json, sys
d = json.load(open(sys.argv[1])); o = json.load(open(sys.argv[2]))
old = {k["kernel"]: k for k in o["kernels"]}
print(d["value"], d["ms_per_step"], "was", o["ms_per_step"])
tot = tot0 = 0
for k in d["kernels"]:
    p = old.get(k["kernel"], {}).get("ms_per_step", 0)
    if "mlp_" in k["kernel"]:
        tot += k["ms_per_step"]; tot0 += p
    if k["ms_per_step"] > 0.1:
        print("%-56s ms=%7.3f was %7.3f" % (k["kernel"][:56], k["ms_per_step"], p))
print("mlp total %.3f was %.3f" % (tot, tot0))
