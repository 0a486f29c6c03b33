#!/usr/bin/env python
"""Ordered kernel timeline of ONE training step out of a rocprofv3 --kernel-trace result (sqlite .db, view
`kernels`): every launch of the last complete step with its duration and the idle gap in front of it.
   python tools/prof_timeline.py <results.db> <out.md> [marker-substring, default adam_kernel]"""
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
    rows = cur.execute('select name, start, end from kernels order by start').fetchall()
    ends = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(ends) < 2:
        print('fewer than two %s launches' % marker)
        sys.exit(1)
    step = rows[ends[-2] + 1:ends[-1] + 1]
    t0 = rows[ends[-2]][2]
    busy = sum(r[2] - r[1] for r in step)
    with open(out, 'w') as f:
        f.write('# kernel timeline of one step (%s): %d launches, %.3f ms wall, %.3f ms busy\n\n' % (
            db.split('/')[-1], len(step), (step[-1][2] - t0) / 1e6, busy / 1e6))
        f.write('| # | kernel | us | gap before, us |\n|---:|---|---:|---:|\n')
        prev = t0
        for i, (name, s, e) in enumerate(step):
            name = name.replace('(anonymous namespace)::', '').replace('|', '/')
            if len(name) > 90:
                name = name[:87] + '...'
            f.write('| %d | `%s` | %.1f | %.1f |\n' % (i, name, (e - s) / 1e3, (s - prev) / 1e3))
            prev = e
    print('wrote', out, len(step), 'launches')


if __name__ == '__main__':
    main()
