#!/usr/bin/env python3
"""Kernel timeline of the decode steps of a rocprofv3 --kernel-trace run of tools/bench_model.py: per kernel name the
mean duration and the mean idle gap in front of it, over the launches of the last decode steps (graph replays).

    python tools/model_timeline.py <run_results.db> [layers]"""
import sqlite3, sys, collections

db = sys.argv[1]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
# the decode steps are the tail of the run: take the last 40 % of the launches
rows = rows[int(len(rows) * 0.6):]
dur = collections.defaultdict(list)
gap = collections.defaultdict(list)
prev_end = None
for name, s, e in rows:
    key = name.replace("void ", "")[:70]
    dur[key].append((e - s) / 1e3)
    if prev_end is not None:
        gap[key].append(max(0.0, (s - prev_end) / 1e3))
    prev_end = max(prev_end or 0, e)
wall = (rows[-1][2] - rows[0][1]) / 1e3
busy = sum(sum(v) for v in dur.values())
print(f"wall {wall:.0f} us, kernel time {busy:.0f} us ({100 * busy / wall:.1f} %), idle {wall - busy:.0f} us over {len(rows)} launches")
print(f"{'kernel':72s} {'n':>6s} {'dur_us':>8s} {'gap_us':>8s} {'tot_ms':>8s} {'gap_ms':>8s}")
for k in sorted(dur, key=lambda k: -sum(dur[k]))[:24]:
    g = gap.get(k, [0.0])
    print(f"{k:72s} {len(dur[k]):6d} {sum(dur[k]) / len(dur[k]):8.2f} {sum(g) / len(g):8.2f} {sum(dur[k]) / 1e3:8.2f} {sum(g) / 1e3:8.2f}")
