"""Print what tools/r03_iter.sh <tag> brought back."""
import json, sys
tag = sys.argv[1]
d = "gpurun_out/r03/"
print("".join(open(d + "det_%s.txt" % tag).readlines()[-3:]), end="")
print(open(d + "t_%s.txt" % tag).readlines()[-1], end="")
print("".join(open(d + "sub_%s.txt" % tag).readlines()[-6:]), end="")
for l in open(d + "bench_%s.jsonl" % tag).readlines()[-3:]:
    j = json.loads(l); print(j["config"]["workload"][:18], "ms/step %.2f" % j["ms_per_step"], "frac %.3f" % j["roofline"]["frac"])
