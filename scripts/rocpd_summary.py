"""Summarise rocprofv3 rocpd (sqlite) outputs into the text files kept under profiles/.

    python scripts/rocpd_summary.py stats <results.db>          # == --kernel-trace --stats summary
    python scripts/rocpd_summary.py pmc <results.db> [<results.db> ...]   # per-kernel mean counter values
"""
import sqlite3
import sys
from collections import defaultdict


def stats(db):
    con = sqlite3.connect(db)
    rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    print(f"# rocprofv3 --kernel-trace --stats  ({db})")
    print(f"{'kernel':<70} {'calls':>6} {'total_us':>12} {'avg_us':>10} {'pct':>6}")
    for name, calls, total, avg, pct in rows:
        print(f"{name[:70]:<70} {calls:>6} {total:>12.1f} {avg:>10.2f} {pct:>6.2f}")
    # per-shape split of templated GEMM kernels: group by (name, grid)
    print("\n# per (kernel, grid) split")
    q = "select name, grid_x, count(*), avg(duration)/1000.0, min(duration)/1000.0, max(duration)/1000.0, vgpr_count, lds_size from kernels group by name, grid_x order by sum(duration) desc"
    print(f"{'kernel':<50} {'grid_x':>9} {'calls':>6} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'vgpr':>5} {'lds':>7}")
    for name, gx, n, avg, mn, mx, vg, lds in con.execute(q):
        print(f"{name[:50]:<50} {gx:>9} {n:>6} {avg:>9.2f} {mn:>9.2f} {mx:>9.2f} {vg:>5} {lds:>7}")


def pmc(dbs):
    agg = defaultdict(lambda: defaultdict(list))
    for db in dbs:
        con = sqlite3.connect(db)
        for name, gx, cname, val in con.execute("select kernel_name, grid_size_x, counter_name, value from counters_collection"):
            agg[(name, gx)][cname].append(val)
    print("# rocprofv3 --kernel-trace --pmc  (mean per dispatch; FETCH_SIZE/WRITE_SIZE in KiB as reported)")
    print("# NOTE gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming reads (MI355X_MICROARCH.md §HBM):")
    print("#      fetch_corrected_MB = 2 * FETCH_SIZE KiB / 1024; WRITE_SIZE is uncalibrated (reported as is)")
    print(f"{'kernel':<50} {'grid':>9} {'n':>5} {'FETCH_KiB':>12} {'fetch_corr_MB':>14} {'WRITE_KiB':>12} {'write_MB':>9}")
    for (name, gx), c in sorted(agg.items(), key=lambda kv: -sum(kv[1].get('FETCH_SIZE', [0]))):
        f = c.get("FETCH_SIZE", [])
        w = c.get("WRITE_SIZE", [])
        fm = sum(f) / len(f) if f else float("nan")
        wm = sum(w) / len(w) if w else float("nan")
        print(f"{name[:50]:<50} {gx:>9} {max(len(f), len(w)):>5} {fm:>12.1f} {2 * fm / 1024:>14.1f} {wm:>12.1f} {wm / 1024:>9.1f}")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2:])
