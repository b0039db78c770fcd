"""Scratch instructions and register counts per FUNCTION of the device code (kernels and noinline phase functions): compiles
builtin_models.hip to assembly and counts scratch_load / scratch_store per symbol, with the LOOP DEPTH (LLVM's annotations) each
sits at — depth 0 = prologue / epilogue (callee-saved registers), 1 = once per pass of the outermost loop, >= 2 = inside a
timestep loop of the small-model kernels.   python tools/scratch_by_function.py [filter]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(ROOT, "iterativelqr.jl_amd", "csrc")
flt = sys.argv[1] if len(sys.argv) > 1 else ""
with tempfile.TemporaryDirectory() as d:
    asm = os.path.join(d, "bm.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-mllvm", "-amdgpu-mfma-vgpr-form", "-S",
                           "--cuda-device-only", "-o", asm, os.path.join(csrc, "builtin_models.hip")], stderr=subprocess.DEVNULL)
    cur, stats, depth = None, {}, 0
    for line in open(asm):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1); stats[cur] = dict(ld=0, st=0, mfma=0)
            continue
        if cur is None:
            continue
        mb = re.match(r"^(\.LBB\d+_\d+|; %bb\.\d+):(.*)$", line)
        if mb:
            depth = 0
            txt = mb.group(2)
            stats[cur]["_pending"] = True
        if stats[cur].get("_pending") and "Loop" in line:
            hh = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", line)
            mm = re.search(r"in Loop: Header=BB\d+_\d+ Depth=(\d+)", line)
            if hh: depth = int(hh.group(1))
            elif mm: depth = int(mm.group(1))
        if line.strip() and not line.strip().startswith(";") and not mb:
            stats[cur]["_pending"] = False
        if "scratch_load" in line or "scratch_store" in line:
            stats[cur]["ld" if "scratch_load" in line else "st"] += 1
            d_ = stats[cur].setdefault("depth", {})
            d_[depth] = d_.get(depth, 0) + 1
        if "v_mfma" in line: stats[cur]["mfma"] += 1
        m = re.match(r"\s*; (NumVgprs|ScratchSize|Occupancy): (\d+)", line)
        if m: stats[cur][m.group(1)] = int(m.group(2))
for k, v in stats.items():
    if flt and flt not in k:
        continue
    if "NumVgprs" not in v:
        continue
    dem = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:100]
    print("%-100s vgpr %3d scratch %5d B  loads %4d stores %4d  mfma %3d  scratch ops by loop depth %s"
          % (dem, v["NumVgprs"], v.get("ScratchSize", 0), v["ld"], v["st"], v["mfma"], dict(sorted(v.get("depth", {}).items()))))
