"""Are a kernel's scratch accesses inside its serial loops? For every serial loop of a kernel (found by its ILQR_ISA_MARK step markers in
the assembly of the library's own compilation) counts the scratch instructions that lie BETWEEN two consecutive step markers of one
loop body — the ones a timestep pays for; everything else is paid per pass or per call.   python tools/scratch_between_steps.py [filter ...]"""
import re, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import issue_model as im
from collections import defaultdict
lines = im.assembly()
funcs = im.functions(lines)
flt = sys.argv[1:] or ["solve_kernel_packedI13Model_acrobotLb0E", "solve_kernel_packedI13Model_acrobotLb1E", "solve_kernel_packedI9Model_carLb1E",
                       "12solve_kernelI13Model_acrobotE", "12solve_kernelI9Model_carE"]
NAMES = {"rollout_step 0": "latency rollout (wave 0)", "riccati_step 1": "latency matrix chain (wave 0)", "riccati_step 2": "latency vector chain (wave 1)",
         "riccati_step 4": "packed Riccati step", "rollout_step 3": "packed rollout step", "delta_step 0": "sensitivity sweep (stage kernels)"}
for tag in flt:
    for nm, a, b in funcs:
        if tag not in nm: continue
        body = lines[a:b + 1]
        ev = []
        for i, l in enumerate(body):
            if "ILQR_MARK" in l: ev.append(("M", l.strip().split("ILQR_MARK")[1].strip()))
            elif "scratch_" in l: ev.append(("S", None))
            elif re.match(r"^\.LBB\d+_\d+:", l) and "Loop Header" in l: ev.append(("H", None))
        res, last = defaultdict(lambda: [0, 0]), {}
        total = sum(1 for e in ev if e[0] == "S")
        for k, e in enumerate(ev):
            if e[0] != "M": continue
            if e[1] in last:
                seg = ev[last[e[1]] + 1:k]
                if not any(x[0] == "H" for x in seg) and not any(x[0] == "M" and x[1] != e[1] for x in seg):
                    res[e[1]][0] += 1; res[e[1]][1] += sum(1 for x in seg if x[0] == "S")
            last[e[1]] = k
        print("%s: %d scratch instructions in all" % (nm[:100], total))
        for m_, (pairs, s) in sorted(res.items()):
            print("    %-34s %2d consecutive step pairs inside one loop body, %3d scratch instructions between them" % (NAMES.get(m_, m_), pairs, s))
