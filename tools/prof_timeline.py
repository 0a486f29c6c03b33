#!/usr/bin/env python
"""Ordered kernel timeline of ONE training step out of a rocprofv3 --kernel-trace result (sqlite .db, view
`kernels`): every launch of the last complete step with its duration and the idle gap in front of it.
   python tools/prof_timeline.py <results.db> <out.md> [marker-substring, default adam_kernel] [step index, default -1]
With a step index (0 = the first step that is preceded by a marker) the chosen step may be one of bench.py's FIRST pass
(no per-kernel events, side streams on): the table then also gives each launch's start offset and queue, and the header the
union-busy time (kernels overlap there).  The step boundary on several queues: all launches that START after the previous
marker launch ended and not after this step's marker launch."""
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    marker = sys.argv[3] if len(sys.argv) > 3 else 'adam_kernel'
    con = sqlite3.connect(db)
    cur = con.cursor()
    names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('view','table')").fetchall()]
    if 'kernels' not in names:
        print('no `kernels` view; objects:', names)
        sys.exit(1)
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    if not {'name', 'start', 'end'} <= set(cols):
        print('unexpected columns:', cols)
        sys.exit(1)
    which = int(sys.argv[4]) if len(sys.argv) > 4 else -1
    qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
    rows = cur.execute('select name, start, end%s from kernels order by start' % (', ' + qcol if qcol else '')).fetchall()
    ends = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(ends) < 2:
        print('fewer than two %s launches' % marker)
        sys.exit(1)
    hi = ends[which] if which < 0 else ends[which + 1]
    lo = ends[ends.index(hi) - 1]
    step = rows[lo + 1:hi + 1]
    t0 = rows[lo][2]
    busy = sum(r[2] - r[1] for r in step)
    union, cur_end = 0, t0
    for r in step:
        s_, e_ = max(r[1], cur_end), r[2]
        if e_ > s_:
            union += e_ - s_
            cur_end = e_
    wall = max(r[2] for r in step) - t0
    queues = sorted(set(r[3] for r in step)) if qcol else []
    with open(out, 'w') as f:
        f.write('# kernel timeline of one step (%s, step %d): %d launches, %.3f ms wall, %.3f ms sum of durations, %.3f ms '
                'with at least one kernel running (idle %.3f ms), %d queue(s)\n\n' % (
                    db.split('/')[-1], which, len(step), wall / 1e6, busy / 1e6, union / 1e6, (wall - union) / 1e6,
                    max(len(queues), 1)))
        f.write('| # | kernel | start, us | us | idle before (nothing running), us | queue |\n|---:|---|---:|---:|---:|---:|\n')
        frontier = t0
        for i, r in enumerate(step):
            name, s, e = r[0], r[1], r[2]
            name = name.replace('(anonymous namespace)::', '').replace('|', '/')
            if len(name) > 90:
                name = name[:87] + '...'
            f.write('| %d | `%s` | %.1f | %.1f | %.1f | %s |\n' % (i, name, (s - t0) / 1e3, (e - s) / 1e3,
                                                                max(s - frontier, 0) / 1e3,
                                                                queues.index(r[3]) if qcol else ''))
            frontier = max(frontier, e)
    print('wrote', out, len(step), 'launches, wall %.3f ms, union busy %.3f ms' % (wall / 1e6, union / 1e6))


if __name__ == '__main__':
    main()
