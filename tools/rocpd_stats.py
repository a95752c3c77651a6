"""Summarise a rocprofv3 rocpd SQLite database (the default --kernel-trace output on ROCm 7.2) into a
per-kernel table: calls, total ms, %, avg / min / max us.  `--last-ms W` restricts to the final W ms of
the trace (e.g. the last timed iteration of bench.py, excluding MIOpen's first-call find phase).

    python tools/rocpd_stats.py gpurun_out/prof/bench_results.db [--last-ms 377] [--top 40] > profiles/x.txt
"""
import argparse
import sqlite3


def from_csv(path, last_ms, top, match):
    """rocprofv3 --kernel-trace --output-format csv  ->  same table."""
    import csv
    rows = list(csv.DictReader(open(path)))
    t1 = max(int(r['End_Timestamp']) for r in rows)
    t0 = min(int(r['Start_Timestamp']) for r in rows)
    lo = t0 if last_ms is None else t1 - int(last_ms * 1e6)
    agg = {}
    for r in rows:
        if int(r['Start_Timestamp']) < lo:
            continue
        d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        a = agg.setdefault(r['Kernel_Name'], [0, 0, 1 << 62, 0, r.get('VGPR_Count', ''), r.get('LDS_Block_Size', '')])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print("# %s  window %.1f ms  kernel time %.2f ms  dispatches %d" % (path, (t1 - lo) / 1e6, tot / 1e6, sum(a[0] for a in agg.values())))
    print("%10s %6s %7s %10s %10s %10s %5s %7s  %s" % ("total_ms", "pct", "calls", "avg_us", "min_us", "max_us", "vgpr", "lds", "kernel"))
    shown = 0
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if match and match not in name:
            continue
        print("%10.3f %6.2f %7d %10.1f %10.1f %10.1f %5s %7s  %s" % (a[1] / 1e6, 100.0 * a[1] / tot, a[0], a[1] / a[0] / 1e3,
                                                                   a[2] / 1e3, a[3] / 1e3, a[4], a[5], name[:150]))
        shown += 1
        if shown >= top:
            break


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--last-ms", type=float, default=None)
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--match", default=None, help="only kernels whose name contains this")
    o = ap.parse_args()
    if o.db.endswith('.csv'):
        return from_csv(o.db, o.last_ms, o.top, o.match)
    c = sqlite3.connect(o.db)
    t1 = c.execute("select max(end) from rocpd_kernel_dispatch").fetchone()[0]
    t0 = c.execute("select min(start) from rocpd_kernel_dispatch").fetchone()[0]
    lo = t0 if o.last_ms is None else t1 - int(o.last_ms * 1e6)
    rows = c.execute(
        "select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start),"
        " max(s.arch_vgpr_count), max(d.group_segment_size), max(d.workgroup_size_x), max(d.grid_size_x)"
        " from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id"
        " where d.start >= ? group by s.kernel_name order by 3 desc", (lo,)).fetchall()
    tot = sum(r[2] for r in rows)
    print("# %s  window %.1f ms  kernel time %.2f ms  dispatches %d" % (o.db, (t1 - lo) / 1e6, tot / 1e6, sum(r[1] for r in rows)))
    print("%10s %6s %7s %10s %10s %10s %5s %7s  %s" % ("total_ms", "pct", "calls", "avg_us", "min_us", "max_us", "vgpr", "lds", "kernel"))
    shown = 0
    for r in rows:
        if o.match and o.match not in r[0]:
            continue
        print("%10.3f %6.2f %7d %10.1f %10.1f %10.1f %5s %7s  %s" % (r[2] / 1e6, 100.0 * r[2] / tot, r[1], r[3] / 1e3, r[4] / 1e3,
                                                                   r[5] / 1e3, r[6], r[7], r[0][:150]))
        shown += 1
        if shown >= o.top:
            break


if __name__ == "__main__":
    main()
