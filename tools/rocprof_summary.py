#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.x rocpd sqlite) output into small text tables for profiles/.

    python tools/rocprof_summary.py stats  <run_results.db>            # kernel-trace --stats
    python tools/rocprof_summary.py pmc    <run_results.db> [filter]   # --pmc passes (per-kernel means)
"""
import sqlite3
import sys

SETUP = " [set-up: cache placement probes, before the first prefix pass]"


def short(name, n=90):
    name = name.replace("void ", "")
    tail = ""
    if name.endswith(SETUP):
        name, tail = name[: -len(SETUP)], " [set-up: placement probes]"
    n -= len(tail)
    return (name if len(name) <= n else name[: n - 3] + "...") + tail


def _label(cur, table, col):
    """SQL expression naming a launch: suffix-pass launches that START before the process's first prefix-pass launch are the
    placement probes of hydragen_amd/placement.py (they run before anything else touches the caches) and get a row of their own,
    so that the suffix pass's count and average are those of the bench's schedule."""
    t0 = cur.execute(f"select min(start) from {table} where {col} like '%prefix_attn%'").fetchone()[0]
    if t0 is None:
        return col
    return f"case when start < {int(t0)} and {col} like '%suffix_attn%' then {col} || '{SETUP}' else {col} end"


def stats(db):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute(
        f"select {_label(cur, 'kernels', 'name')} as nm, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by nm order by 3 desc"
    ).fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"{'kernel':92s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for name, n, tot, avg, mn, mx in rows[:12]:
        print(f"{short(name):92s} {n:6d} {tot/1e3:12.1f} {avg/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*tot/total:6.2f}")


def pmc(db, filt=None):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute(
        f"select {_label(cur, 'counters_collection', 'kernel_name')} as nm, counter_name, count(*), avg(value), sum(value), avg(duration) "
        "from counters_collection group by nm, counter_name order by 1, 2"
    ).fetchall()
    print(f"{'kernel':72s} {'counter':28s} {'n':>5s} {'mean':>16s} {'sum':>18s} {'avg_dur_us':>11s}")
    for name, cname, n, avg, tot, dur in rows:
        if filt and filt not in name:
            continue
        print(f"{short(name, 72):72s} {cname:28s} {n:5d} {avg:16.1f} {tot:18.1f} {dur/1e3:11.2f}")


if __name__ == "__main__":
    mode, db = sys.argv[1], sys.argv[2]
    if mode == "stats":
        stats(db)
    else:
        pmc(db, sys.argv[3] if len(sys.argv) > 3 else None)
