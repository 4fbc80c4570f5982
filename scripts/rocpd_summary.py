#!/usr/bin/env python3
"""rocprofv3 rocpd database (or its csv output dir) -> a small text summary for profiles/.

    python scripts/rocpd_summary.py gpurun_out/prof_r1/x_results.db > profiles/r01_x_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    print("# per-kernel durations from %s (rocprofv3 --kernel-trace --stats)" % path.split("/")[-1])
    rows = list(db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(grid_x), "
        "max(workgroup_x), max(lds_size), max(vgpr_count), max(sgpr_count) from kernels group by name "
        "order by sum(duration) desc"))
    tot = sum(r[2] for r in rows) or 1
    print("%-52s %5s %13s %13s %13s %13s %6s %8s %3s %6s %5s %5s" % (
        "kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "%", "grid", "wg", "lds", "vgpr", "sgpr"))
    for name, calls, total, avg, mn, mx, grid, wg, lds, vg, sg in rows:
        print("%-52s %5d %13d %13.0f %13d %13d %6.2f %8d %3d %6d %5d %5d" % (
            name[:52], calls, total, avg, mn, mx, 100.0 * total / tot, grid, wg, lds, vg, sg))
    try:
        rows = list(db.execute(
            "select k.kernel_name, p.counter_name, sum(p.value), count(*) from pmc_events p "
            "join kernels k on p.event_id = k.event_id group by k.kernel_name, p.counter_name"))
    except sqlite3.Error:
        rows = []
    if rows:
        print("\n# PMC counters summed over dispatches")
        print("%-58s %-16s %18s %6s" % ("kernel", "counter", "sum", "n"))
        for k, c, v, n in rows:
            print("%-58s %-16s %18.1f %6d" % (k[:58], c, v, n))


if __name__ == "__main__":
    main(sys.argv[1])
