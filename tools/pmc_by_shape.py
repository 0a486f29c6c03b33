#!/usr/bin/env python
"""FETCH_SIZE / WRITE_SIZE per GEMM SHAPE from a rocprofv3 --pmc pass over `tools/gemm_split_probe.py one <lib> <shapes>`:
the probe issues 23 launches per shape in the order given, so the dispatches of the main kernels are attributed by order.
   python tools/pmc_by_shape.py <results.db> <counter> <shape> [<shape> ...]
Prints the average counter value per launch in MB (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes) next to the
operand bytes of the shape."""
import sqlite3
import sys


def main():
    db, counter = sys.argv[1], sys.argv[2]
    shapes = sys.argv[3:]
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute('pragma table_info(counters_collection)').fetchall()]
    order = 'dispatch_id' if 'dispatch_id' in cols else ('start' if 'start' in cols else 'id')
    rows = cur.execute('select kernel_name, sum(value), %s from counters_collection where counter_name = ? group by %s '
                       'order by %s' % (order, order, order), (counter,)).fetchall()
    main_rows = [r for r in rows if 'gemm_split' in r[0] and 'reduce' not in r[0]]
    red_rows = [r for r in rows if 'split_reduce' in r[0]]
    per = 23
    print('%d main dispatches, %d reduce dispatches, %d shapes' % (len(main_rows), len(red_rows), len(shapes)))
    scale = 2.0 if counter == 'FETCH_SIZE' else 1.0
    i = 0
    for spec in shapes:
        v = [int(x) for x in spec.split(',')]
        m, n, k = v[:3]
        sk = v[5] if len(v) > 5 else 1
        chunk = main_rows[i:i + per]
        i += per
        if not chunk:
            break
        avg = sum(r[1] for r in chunk[3:]) / max(len(chunk[3:]), 1) * 1024 * scale / 1e6
        reads = (m * k + n * k) * 4 / 1e6
        writes = m * n * 4 * (sk if sk > 1 else 1) / 1e6
        print('%-28s %-36s %s x%g = %8.1f MB per launch   (operands read once %.1f MB, output / partials %.1f MB)' % (
            spec, chunk[-1][0].replace('(anonymous namespace)::', '').split('(')[0][-36:], counter, scale, avg, reads, writes))


if __name__ == '__main__':
    main()
