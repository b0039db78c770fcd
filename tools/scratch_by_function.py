"""Scratch instructions and register counts per FUNCTION of the device code (kernels and noinline phase functions): compiles
builtin_models.hip to assembly and counts scratch_load / scratch_store per symbol.  python tools/scratch_by_function.py [filter]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(ROOT, "iterativelqr.jl_amd", "csrc")
flt = sys.argv[1] if len(sys.argv) > 1 else ""
with tempfile.TemporaryDirectory() as d:
    asm = os.path.join(d, "bm.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-mllvm", "-amdgpu-mfma-vgpr-form", "-S",
                           "--cuda-device-only", "-o", asm, os.path.join(csrc, "builtin_models.hip")], stderr=subprocess.DEVNULL)
    cur, stats = None, {}
    for line in open(asm):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1); stats[cur] = dict(ld=0, st=0, mfma=0)
            continue
        if cur is None:
            continue
        if "scratch_load" in line: stats[cur]["ld"] += 1
        if "scratch_store" in line: stats[cur]["st"] += 1
        if "v_mfma" in line: stats[cur]["mfma"] += 1
        m = re.match(r"\s*; (NumVgprs|ScratchSize|Occupancy): (\d+)", line)
        if m: stats[cur][m.group(1)] = int(m.group(2))
for k, v in stats.items():
    if flt and flt not in k:
        continue
    if "NumVgprs" not in v:
        continue
    dem = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:100]
    print("%-100s vgpr %3d scratch %5d B  loads %4d stores %4d  mfma %3d" % (dem, v["NumVgprs"], v.get("ScratchSize", 0), v["ld"], v["st"], v["mfma"]))
