"""Instruction-issue model of the latency kernel's critical path, from the ISA of THIS build.

    python tools/issue_model.py --asm FILE.s --out issue_model.json [--flags "..."]    # what csrc/Makefile runs on the assembly
                                                                                        # the library's own compilation kept
    python tools/issue_model.py [config ...]      # stand-alone: compiles to assembly itself, prints the loop table

The build writes lib/issue_model.json stamped with a hash of the device sources; bench.py recomputes that hash and drops a
stale model instead of comparing the measurement with another build's instruction lists.

What bounds solve_kernel<Model_acrobot> at batch 1024 is not HBM (counter traffic well under 1 % of peak) but the serial
instruction stream of each instance's slowest wave. tools/probes/probe_issue.hip (profiles/r02_probe_issue.txt) measures what
ONE wave can issue on gfx950: one instruction every ~5.1-6 shader clocks whatever its class (fp64 / integer VALU, DPP,
v_readlane, scalar ALU, s_nop) and whether or not it depends on the previous one; a 4x4x4 f64 MFMA every ~17 clk; an LDS write
every ~13 clk; LDS reads ~7-10 clk. So the step of a serial loop lasts as long as its instruction list, and instruction-level
parallelism inside the wave buys nothing. This script reads the assembly of csrc/builtin_models.hip, finds the serial time
loops of solve_kernel<Model_X> through the comment markers the kernels carry (ILQR_ISA_MARK) and LLVM's loop annotations, and counts the
instructions ISSUED per timestep on each wave by class (s_nop N counts N+1 idle states). bench.py turns them into the issue
time of the slowest instance's critical wave (roofline.issue_model.predicted_floor_ms, achieved_over_floor)."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "iterativelqr.jl_amd", "csrc")
MANGLED = {"acrobot": "_ZN4ilqr12solve_kernelI13Model_acrobotEEvNS_5KArgsE",
           "car": "_ZN4ilqr12solve_kernelI9Model_carEEvNS_5KArgsE"}
HORIZON_UNROLL = 2          # both time loops are unrolled by two (ping-pong operand sets)
# Issue time of one instruction of a LONE wave, in shader clocks (tools/probes/probe_issue.hip, profiles/r02_probe_issue.txt;
# the critical wave shares its SIMD with one other wave, which the probe shows to cost nothing until both are fp64-dense):
# VALU of any kind 5.1-6.0, scalar ALU / s_nop 5.1-5.3 (+1 per extra idle state), v_mfma_f64_4x4x4 17, ds_write_b64 13,
# ds_read_b64 7-10, global load ~6 (issue only)
OCC = {"valu_f64": 5.5, "mfma": 17.0, "valu_other": 5.1, "dpp_perm": 5.5, "lds": 10.0, "vmem": 6.0, "salu": 5.2, "waitcnt": 5.0, "nop_states": 1.0}
CLOCK_GHZ = 2.38            # sustained shader clock with 1024 such workgroups resident (probe_clock.hip)


def device_source_hash():
    """sha256 over the device sources a build of the library is made from (names and contents, sorted)."""
    import hashlib
    files = sorted([os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp") or f == "builtin_models.hip"] +
                   [os.path.join(CSRC, "models", f) for f in os.listdir(os.path.join(CSRC, "models")) if f.endswith(".h")] +
                   [os.path.join(ROOT, "include", "ilqr_hip.h")])
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def assembly():
    out = "/tmp/ilqr_builtin_models.s"
    src = os.path.join(CSRC, "builtin_models.hip")
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(CSRC, "models", f) for f in os.listdir(os.path.join(CSRC, "models"))]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-mllvm", "-amdgpu-mfma-vgpr-form",
                               "--cuda-device-only", "-S", "-I", CSRC, src, "-o", out], stderr=subprocess.DEVNULL)
    return open(out).read().splitlines()


def function_body(lines, mangled):
    start = next(i for i, ln in enumerate(lines) if ln.startswith(mangled + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    return lines[start:end + 1]


def loops(body):
    """(first, last) line indices of innermost loops: a branch to a label defined above it, with no other loop inside."""
    labels = {}
    for i, ln in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            labels[m.group(1)] = i
    spans = []
    for i, ln in enumerate(body):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", ln) or re.match(r"\s+s_branch\s+(\.LBB\d+_\d+)", ln)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            spans.append((labels[m.group(1)], i))
    inner = [s for s in spans if not any(o != s and s[0] <= o[0] and o[1] <= s[1] for o in spans)]
    return inner


def classify(body, span):
    c = {"total": 0, "valu_f64": 0, "valu_other": 0, "mfma": 0, "salu": 0, "lds": 0, "vmem": 0, "nop_states": 0, "waitcnt": 0, "dpp_perm": 0}
    for ln in body[span[0]:span[1] + 1]:
        t = ln.strip()
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        op = t.split()[0]
        if op == "s_nop":
            c["nop_states"] += int(t.split()[1]) + 1
            c["total"] += 1
            continue
        c["total"] += 1
        if op.startswith("v_mfma"):
            c["mfma"] += 1
        elif op.startswith("v_") and ("f64" in op or op in ("v_div_scale_f64", "v_div_fmas_f64", "v_div_fixup_f64")):
            c["valu_f64"] += 1
        elif op.startswith("v_") and ("dpp" in t or "permlane" in op or "readlane" in op or "readfirstlane" in op):
            c["dpp_perm"] += 1
        elif op.startswith("v_"):
            c["valu_other"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            c["vmem"] += 1
        elif op == "s_waitcnt":
            c["waitcnt"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
    return c


def block_loops(body):
    """Per line: (loop header id, depth) of the basic block it belongs to, from LLVM's own loop annotations
    ('; in Loop: Header=BBf_h Depth=d' on member blocks, '; =>This [Inner] Loop Header: Depth=d' on headers)."""
    owner = [None] * len(body)
    cur = None
    i = 0
    while i < len(body):
        ln = body[i]
        m = re.match(r"^(?:\.LBB\d+_(\d+)|; %bb\.(\d+)):\s*(?:;(.*))?$", ln)
        if m:
            bid = m.group(1) or m.group(2)
            cm = m.group(3) or ""
            j = i + 1
            while j < len(body) and re.match(r"^\s+;", body[j]) and "Loop" in body[j]:
                cm += " " + body[j]
                j += 1
            mm = re.search(r"in Loop: Header=BB\d+_(\d+) Depth=(\d+)", cm)
            hh = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", cm)
            if hh:
                cur = (bid, int(hh.group(1)))
            elif mm:
                cur = (mm.group(1), int(mm.group(2)))
            else:
                cur = None
        owner[i] = cur
        i += 1
    return owner


def marked_loop(body, marker):
    """The deepest loop that holds `marker` comments: instruction counts of its own blocks (child loops — cold slow
    paths such as the huge-argument trig reduction — excluded) and the number of step copies in its body."""
    owner = block_loops(body)
    marks = [i for i, ln in enumerate(body) if marker in ln and owner[i] is not None]
    if not marks:
        raise RuntimeError("marker %r not inside any loop" % marker)
    loop = max((owner[i] for i in marks), key=lambda o: o[1])
    k = sum(1 for i in marks if owner[i] == loop)
    lines = [body[i] for i in range(len(body)) if owner[i] == loop]
    # rarely taken straight-line regions (ILQR_ISA_COLD_BEGIN / _END in the kernels: the repeat of a factorisation after a failed
    # pivot) are not part of the step's list either; the compiler may place such a block anywhere inside the loop's blocks
    kept, cold = [], 0
    for ln in lines:
        if "ILQR_COLD_BEGIN" in ln:
            cold += 1
        elif "ILQR_COLD_END" in ln:
            cold = max(0, cold - 1)
        elif cold == 0:
            kept.append(ln)
    lines = kept
    c = classify(lines, (0, len(lines) - 1))
    c["chain"] = recurrence_chain(lines)
    return c, k


# ---- dependent-chain analysis -------------------------------------------------------------------------------------------
# Latency of one instruction as seen by a DEPENDENT successor of the same wave, in shader clocks. VALU / SALU / DPP: the
# dependent issue interval of tools/probes/probe_issue.hip (a dependent instruction issues as soon as an independent one would);
# v_mfma_f64_4x4x4: 19 (result as the B operand of the next), 17 as its C operand — 19 taken; LDS read ~64 and write 13 alone on
# the chip (profiles/r02_probe_lds.txt); a global load answered by the L2 ~500 (operands are requested ahead, so they are on the
# recurrence only through their address registers).
CHAIN_LAT = {"valu_f64": 5.5, "mfma": 19.0, "valu_other": 5.1, "dpp_perm": 5.5, "lds": 64.0, "vmem": 500.0, "salu": 5.2, "waitcnt": 0.0, "nop": 0.0}
_NO_DST = ("s_cmp", "s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_barrier", "s_bitcmp", "s_setprio", "s_sleep", "s_endpgm",
           "ds_write", "global_store", "scratch_store", "flat_store", "buffer_store", "s_setreg", "s_sendmsg")


def _regs(tok):
    """registers named by one operand token: v12, v[4:7], s3, s[10:11], vcc, exec, -v[1:2], |v[1:2]|, ..."""
    tok = tok.strip().strip("-|").strip()
    m = re.match(r"^([vsa])\[(\d+):(\d+)\]$", tok)
    if m:
        return ["%s%d" % (m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)]
    m = re.match(r"^([vsa])(\d+)$", tok)
    if m:
        return [tok]
    if tok in ("vcc", "vcc_lo", "vcc_hi"):
        return ["vcc"]
    if tok in ("exec", "exec_lo", "exec_hi"):
        return ["exec"]
    if tok == "m0":
        return ["m0"]
    return []


def _instr_class(t):
    op = t.split()[0]
    if op == "s_nop":
        return "nop"
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_") and ("f64" in op or op in ("v_div_scale_f64", "v_div_fmas_f64", "v_div_fixup_f64")) and "dpp" not in t:
        return "valu_f64"
    if op.startswith("v_") and ("dpp" in t or "permlane" in op or "readlane" in op or "readfirstlane" in op):
        return "dpp_perm"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op == "s_waitcnt":
        return "waitcnt"
    return "salu"


def _defs_uses(t):
    op = t.split()[0]
    rest = t[len(op):]
    rest = re.sub(r"\b(offset\d*|row_\w+|quad_perm|bank_mask|row_mask|bound_ctrl|bitop3|cbsz|abid|blgp|op_sel\w*|neg_\w+|clamp|glc|slc|sc0|sc1|nt)[:\[\w,\]]*", "", rest)
    toks = [x for x in rest.split(",")]
    ops = [_regs(x.split()[0]) if x.split() else [] for x in toks]
    defs, uses = [], []
    no_dst = op.startswith(_NO_DST)
    if op.startswith("v_cmp") and op.endswith("_e32"):
        defs = ["vcc"]; uses = [r for o in ops for r in o]
    elif no_dst:
        uses = [r for o in ops for r in o]
    else:
        defs = ops[0] if ops else []
        uses = [r for o in ops[1:] for r in o]
        if op.startswith(("v_fmac", "v_mac", "v_dot")) or "dpp" in t:
            uses += defs                                  # two-address forms (and DPP keeps the old value of masked lanes)
    if op.startswith(("v_cndmask_b32_e32", "v_addc", "v_subb", "v_div_fmas")):
        uses.append("vcc")
    if op.startswith("s_") and not no_dst and not op.startswith(("s_mov", "s_cselect", "s_load", "s_getreg", "s_movk")):
        defs = defs + ["scc"]
    if op.startswith(("s_cmp", "s_bitcmp")):
        defs = ["scc"]
    if op.startswith(("s_cselect", "s_cbranch_scc", "s_addc", "s_subb")):
        uses.append("scc")
    if op.startswith("s_cbranch_vcc"):
        uses.append("vcc")
    if op.endswith("saveexec_b64"):
        defs = defs + ["exec"]; uses.append("exec")
    return defs, uses


def recurrence_chain(lines, copies=4):
    """Longest chain of true (read-after-write) register dependences that one trip of the loop adds, in instructions and in
    clocks at CHAIN_LAT: the loop's own instruction list (program order, cold child loops excluded) is laid out `copies` times
    and the longest path through copies 1..k is taken; the growth from k - 1 to k copies is the recurrence of one trip — what
    bounds the loop however many instructions could issue beside it. Memory dependences through LDS are not followed (operands
    are requested one or two steps ahead; the hand-over between the two waves is a barrier)."""
    ins = []
    for ln in lines:
        t = ln.strip()
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        t = t.split(";")[0].strip()
        if not t:
            continue
        cls = _instr_class(t)
        d, u = _defs_uses(t)
        w = CHAIN_LAT[cls]
        if cls == "nop":
            w = 1.0 + int(t.split()[1])
        ins.append((cls, d, u, w))
    best_n, best_c = [], []
    last = {}          # register -> (chain length in instructions, in clocks) of its latest writer
    for k in range(copies):
        for cls, d, u, w in ins:
            n0 = max([last[r][0] for r in u if r in last] or [0])
            c0 = max([last[r][1] for r in u if r in last] or [0.0])
            cnt = 0 if cls in ("nop", "waitcnt") else 1
            for r in d:
                last[r] = (n0 + cnt, c0 + w)
        best_n.append(max(v[0] for v in last.values()))
        best_c.append(max(v[1] for v in last.values()))
    return {"instructions_per_trip": best_n[-1] - best_n[-2], "clk_per_trip": best_c[-1] - best_c[-2]}


def functions(lines):
    """[(name, first, last)] of every function in the assembly (LLVM prints 'name:   ; @name' at a function's entry)."""
    starts = [(i, m.group(1)) for i, ln in enumerate(lines) for m in [re.match(r"^([.\w$]+):\s+; @", ln)] if m]
    return [(nm, i, (starts[k + 1][0] if k + 1 < len(starts) else len(lines)) - 1) for k, (i, nm) in enumerate(starts)]


def body_with_marker(lines, funcs, marker, config):
    """The function of solve_kernel<Model_config>'s call tree that holds `marker`: the kernel itself, or a device function it
    calls (forward_pass, backward_pass_split<.., false, ROLE> are real calls with their own register allocation)."""
    model = "Model_%s" % config
    best = None
    for nm, a, b in funcs:
        if ("%d%s" % (len(model), model)) not in nm or any(x in nm for x in ("Slim", "packed", "stage_kernel", "slim")):
            continue
        if "backward_pass_split" in nm and "Lb0E" not in nm:           # the fused solve's instantiation (STORE_VALUE = false)
            continue
        if any(marker in ln for ln in lines[a:b + 1]):
            if best is None or nm == MANGLED[config]:
                best = (nm, a, b)
    if best is None:
        raise RuntimeError("marker %r not found for %s" % (marker, config))
    return lines[best[1]:best[2] + 1]


def model_for(config, lines):
    funcs = functions(lines)
    mk = lambda marker: marked_loop(body_with_marker(lines, funcs, marker, config), marker)
    rol, krol = mk("ILQR_MARK rollout_step 0")          # wave 0's rollout (two-wave kernel: no MFMA ride-along)
    ric, kric = mk("ILQR_MARK riccati_step 1")          # ROLE 1: matrix chain on wave 0
    vec, kvec = mk("ILQR_MARK riccati_step 2")          # ROLE 2: vector chain on wave 1
    dlt, kdlt = mk("ILQR_MARK delta_step 0")            # wave 1, beside the first rollout
    table = [dict(rol, kind="rollout (wave 0)", steps=krol), dict(ric, kind="riccati matrix chain (wave 0)", steps=kric),
             dict(vec, kind="riccati vector chain (wave 1)", steps=kvec), dict(dlt, kind="delta sweep (wave 1)", steps=kdlt)]
    slots = lambda c, k: (c["total"] + c["nop_states"]) / k
    # every s_nop is an instruction of its own (5 clk at the scalar rate) plus one clock per idle state
    nops = lambda c: c["total"] - sum(c[key] for key in ("valu_f64", "valu_other", "mfma", "salu", "lds", "vmem", "waitcnt", "dpp_perm"))
    occ = lambda c, k: (sum(OCC[key] * c[key] for key in OCC) + 4.2 * nops(c)) / k
    return {
        "kernel": "solve_kernel<Model_%s>, wave 0 (the instance's critical wave)" % config,
        "rollout_step_instr": slots(rol, krol), "riccati_step_instr": slots(ric, kric),
        "wave1_riccati_step_instr": slots(vec, kvec), "wave1_delta_step_instr": slots(dlt, kdlt),
        "rollout_loop": dict(rol, steps_in_body=krol), "riccati_loop": dict(ric, steps_in_body=kric),
        "rollout_step_occupancy_clk": occ(rol, krol), "riccati_step_occupancy_clk": occ(ric, kric),
        "wave1_riccati_step_occupancy_clk": occ(vec, kvec),
        # longest chain of true register dependences per timestep (recurrence_chain): what bounds a step however its other
        # instructions are scheduled; slots / chain says how far the step is from its own critical path
        "rollout_step_chain": {"instructions": rol["chain"]["instructions_per_trip"] / krol, "clk": rol["chain"]["clk_per_trip"] / krol},
        "riccati_step_chain": {"instructions": ric["chain"]["instructions_per_trip"] / kric, "clk": ric["chain"]["clk_per_trip"] / kric},
        "wave1_riccati_step_chain": {"instructions": vec["chain"]["instructions_per_trip"] / kvec, "clk": vec["chain"]["clk_per_trip"] / kvec},
        "rollout_slots_over_chain": slots(rol, krol) / max(1.0, rol["chain"]["instructions_per_trip"] / krol),
        "riccati_slots_over_chain": slots(ric, kric) / max(1.0, ric["chain"]["instructions_per_trip"] / kric),
        "chain_latency_clk_per_instruction": CHAIN_LAT,
        "occupancy_clk_per_instruction": OCC, "clock_ghz": CLOCK_GHZ,
        "note": "issue slots per timestep = (instructions + s_nop idle states of the loop holding the step marker, cold child loops "
                "excluded) / step copies in that loop body (assembly kept by the library's own compilation). *_occupancy_clk = "
                "the issue time of that instruction list at the single-wave rates of tools/probes/probe_issue.hip "
                "(occupancy_clk_per_instruction; s_nop N = 5 + N). A lone wave issues one instruction per 5-6 clk whether or not it "
                "depends on the previous one, so this IS the speed limit of the serial loops; what the measured time adds on top "
                "is fp64-pipe sharing with the other wave of the SIMD, LDS / HBM waits and barriers. *_chain = longest chain of true "
                "register dependences one timestep adds (recurrence_chain): the step's own critical path; slots_over_chain says how "
                "much of the step is issue serialisation of work that does not depend on each other",
    }, table


def main():
    argv = sys.argv[1:]
    opt = {}
    while argv and argv[0] in ("--asm", "--out", "--flags"):
        opt[argv[0][2:]] = argv[1]
        argv = argv[2:]
    lines = open(opt["asm"]).read().splitlines() if "asm" in opt else assembly()
    out = {}
    for config in (argv or ["acrobot", "car"]):
        m, table = model_for(config, lines)
        out[config] = m
        if "asm" in opt:
            continue
        print("== %s: serial loops of the solve kernel (per loop body)" % config)
        for t in table:
            print("  %-32s steps/body %d  total %4d  f64 %4d  mfma %3d  valu %3d  dpp/perm %3d  lds %3d  vmem %3d  salu %3d  wait %2d  nop-states %3d"
                  % (t["kind"], t["steps"], t["total"], t["valu_f64"], t["mfma"], t["valu_other"], t["dpp_perm"], t["lds"], t["vmem"], t["salu"],
                     t["waitcnt"], t["nop_states"]))
        print("  -> wave 0 per timestep: rollout %.0f issue slots = %.0f clk of SIMD occupancy, Riccati matrix chain %.0f issue slots = %.0f clk"
              % (m["rollout_step_instr"], m["rollout_step_occupancy_clk"], m["riccati_step_instr"], m["riccati_step_occupancy_clk"]))
    out["_build"] = {"source_hash": device_source_hash(), "flags": opt.get("flags", ""),
                     "from": "assembly kept by the library's own compilation (-save-temps)" if "asm" in opt else "stand-alone compile of the same sources"}
    json.dump(out, open(opt.get("out", os.path.join(ROOT, "profiles", "issue_model.json")), "w"), indent=1)


if __name__ == "__main__":
    main()
