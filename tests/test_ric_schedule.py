"""The static schedule of the large-model Riccati step (csrc/ilqr_ric_schedule.hpp) is plain constexpr C++: compiled here with
g++ and checked on the host — every tile of ûx, T, Qux, Quu, Qxx, P formed exactly once for nx up to 64 (TN = 1..4), in a window
where its operands exist, every Qxx tile behind the T tiles it reads (own wave or flag), no task list overflowing. The same header
holds the rule by which the workgroups of a CU pick their critical waves (role_mask): optimal and the same for
every reader on all 22 620 placements of one to four workgroups."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_riccati_schedule_invariants(tmp_path):
    exe = str(tmp_path / "ric_schedule_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "iterativelqr.jl_amd", "csrc"),
                           os.path.join(ROOT, "tests", "ric_schedule_check.cpp"), "-o", exe])
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert out.returncode == 0, out.stderr.decode()
    lines = out.stdout.decode().splitlines()
    assert lines[-1] == "roles: %d placements checked" % (12 + 12 ** 2 + 12 ** 3 + 12 ** 4)      # the role rule of DESIGN.md 3.0 (role_mask)
    lines = lines[:-1]
    assert len(lines) == 4 and lines[1].startswith("TN=2: 4 T, 4 Qxx, 4 P tiles") and lines[1].endswith("per wave 3 2 2")
