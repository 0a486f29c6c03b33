#!/usr/bin/env python
"""Turns a rocprofv3 `--kernel-trace --stats` result (sqlite .db, view `top_kernels`) into the small
text summary committed under profiles/.   python tools/prof_summary.py <results.db> <out.md> [steps]"""
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else None
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute('select name, total_calls, total_duration, average, percentage from top_kernels').fetchall()
    total = sum(r[2] for r in rows)
    with open(out, 'w') as f:
        f.write('# rocprofv3 --kernel-trace --stats summary (%s)\n\n' % db.split('/')[-1])
        f.write('total kernel time %.3f ms over %d kernels' % (total / 1e3, len(rows)))
        if steps:
            f.write('; %d profiled steps => %.3f ms of kernel time per step' % (steps, total / 1e3 / steps))
        f.write('\n\n| kernel | calls | total us | avg us | %% |\n|---|---:|---:|---:|---:|\n')
        for name, calls, tot, avg, pct in rows:
            name = name.replace('(anonymous namespace)::', '').replace('|', '/')
            if len(name) > 110:
                name = name[:107] + '...'
            f.write('| `%s` | %d | %.1f | %.2f | %.2f |\n' % (name, calls, tot, avg, pct))


if __name__ == '__main__':
    main()
