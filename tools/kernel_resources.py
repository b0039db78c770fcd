"""VGPR / SGPR / scratch / LDS of every kernel in a built object (metadata notes of the gfx950 code object).
  python tools/kernel_resources.py [object.o] [name filter]      default object: lib/builtin_models.o"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "iterativelqr.jl_amd", "lib", "builtin_models.o")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
LLVM = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as d:
    co = os.path.join(d, "dev.co")
    # the fat binary sits in the .hip_fatbin section of the shared object
    fb = os.path.join(d, "fatbin")
    subprocess.check_call([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fb, lib])
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fb,
                           "--output=" + co, "--unbundle"])
    notes = subprocess.check_output([LLVM + "/llvm-readelf", "--notes", co], text=True)
cur = {}
rows = []
for line in notes.splitlines():
    m = re.match(r"\s*[- ]\s*\.(\w+):\s*(.*)$", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2).strip()
    if k == "agpr_count" and cur:
        rows.append(cur); cur = {}
    cur[k] = v
rows.append(cur)
for r in rows:
    nm = r.get("name", "")
    if not nm or (flt and flt not in nm):
        continue
    dem = subprocess.run(["c++filt", nm], capture_output=True, text=True).stdout.strip()
    print("%-90s vgpr %4s agpr %4s sgpr %4s scratch %6s lds %6s" % (dem[:90], r.get("vgpr_count"), r.get("agpr_count"), r.get("sgpr_count"),
                                                                   r.get("private_segment_fixed_size"), r.get("group_segment_fixed_size")))
