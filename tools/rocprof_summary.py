"""Summarise rocprofv3 (rocpd sqlite) outputs into small text files for profiles/.
Usage: python tools/rocprof_summary.py <kernel_trace.db> [<fetch.db> <write.db>] > profiles/xxx.txt"""
import sqlite3
import sys


def main():
    kt = sqlite3.connect(sys.argv[1]).cursor()
    print("# rocprofv3 --kernel-trace --stats summary (top_kernels view: durations in MICROSECONDS; the per-dispatch avg further down is in ns)")
    print("%-70s %6s %16s %14s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in kt.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print("%-70s %6d %16.0f %14.1f %8.3f" % (name, calls, total, avg, pct))
    print()
    print("# per-dispatch resources of the dominant kernel")
    for r in kt.execute("select name, grid_x, workgroup_x, lds_size, scratch_size, vgpr_count, accum_vgpr_count, sgpr_count, "
                        "avg(duration), count(*) from kernels group by name order by sum(duration) desc limit 1"):
        print(dict(zip(("name", "grid", "workgroup", "lds_bytes", "scratch", "vgpr", "agpr", "sgpr", "avg_ns", "n"), r)))
    for path in sys.argv[2:]:
        cur = sqlite3.connect(path).cursor()
        print()
        print("# PMC pass:", path)
        for name, cname, tot, n in cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection "
                                               "group by kernel_name, counter_name"):
            unit = "KB" if cname.endswith("_SIZE") else "count"
            print("%-62s %-20s dispatches=%d per_dispatch=%.4g %s" % (name, cname, n, tot / n, unit))


if __name__ == "__main__":
    main()
