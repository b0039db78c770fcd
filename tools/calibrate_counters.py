"""FETCH_SIZE / WRITE_SIZE of rocprofv3 against KNOWN traffic in the access patterns of the solve kernels
(tools/probes/probe_traffic.hip): two separate --pmc passes (the guide's rule), counters in KiB, per-kernel averages divided by
the true byte counts. Writes profiles/r03_counter_calibration.{json,txt}; bench.py's measure_traffic applies these factors.
    python tools/calibrate_counters.py            (on the GPU box)"""
import json, os, sqlite3, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(ROOT, "tools", "probes", "probe_traffic")
true = {}
meas = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    with tempfile.TemporaryDirectory(prefix="ilqr_cal_", dir="/tmp") as d:
        p = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", d, "-o", "cal", "--", exe], cwd="/tmp",
                           env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        for ln in p.stdout.decode().splitlines():
            if ln.startswith("TRUE "):
                _, k, r, w = ln.split()
                true[k] = (int(r), int(w))
        dbs = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
        if not dbs:
            sys.exit("no rocprofv3 database for %s:\n%s" % (counter, p.stdout.decode()[-2000:]))
        cur = sqlite3.connect(dbs[0]).cursor()
        for name, tot, n in cur.execute("select kernel_name, sum(value), count(*) from counters_collection where counter_name = ? group by kernel_name", (counter,)):
            key = name.split("(")[0].strip()
            meas.setdefault(key, {})[counter] = tot / n * 1024.0          # KiB -> bytes per launch
out = {}
lines = ["# tools/calibrate_counters.py on MI355X: rocprofv3 counter (KiB -> bytes, per launch) / true bytes of tools/probes/probe_traffic.hip",
         "%-20s %14s %14s %8s %14s %14s %8s" % ("pattern", "true read B", "FETCH_SIZE B", "ratio", "true write B", "WRITE_SIZE B", "ratio")]
for k, (r, w) in true.items():
    m = meas.get(k, {})
    f, ws = m.get("FETCH_SIZE", float("nan")), m.get("WRITE_SIZE", float("nan"))
    out[k] = {"true_read": r, "true_write": w, "fetch_bytes": f, "write_bytes": ws,
              "fetch_over_true": (f / r if r else None), "write_over_true": (ws / w if w else None)}
    lines.append("%-20s %14d %14.0f %8s %14d %14.0f %8s" % (k, r, f, "%.3f" % (f / r) if r else "-", w, ws, "%.3f" % (ws / w) if w else "-"))
os.makedirs(os.path.join(ROOT, "gpurun_out", "r03"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r03", "counter_calibration.json"), "w"), indent=1)
open(os.path.join(ROOT, "gpurun_out", "r03", "counter_calibration.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
