import sys, json
for l in sys.stdin:
    if '"kernel"' not in l:
        continue
    d = json.loads(l)
    print("%-46s %9.3f ms  %6.1f TF/s  %8.1f GB/s" % (d["kernel"], d["ms"], d.get("TFps", 0.0), d["GBps"]))
