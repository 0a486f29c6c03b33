#!/usr/bin/env python
"""Per-kernel averages of SQ counters from one rocprofv3 --pmc pass (sqlite results.db, table counters_collection).
   python tools/pmc_sq.py <results.db> <out.md> [kernel-name substring]
Derived columns: mfma pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES) when both were collected
(MI355X_MICROARCH.md: MFMA_BUSY counts cycles per SIMD; BUSY_CU_CYCLES per CU)."""
import re
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    pat = sys.argv[3] if len(sys.argv) > 3 else ''
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute('select kernel_name, counter_name, count(*), sum(value) from counters_collection '
                       'group by kernel_name, counter_name').fetchall()
    tab, counters = {}, []
    for name, c, n, tot in rows:
        if pat and pat not in name:
            continue
        short = re.sub(r'^void ', '', name.replace('(anonymous namespace)::', ''))
        short = short.split('(')[0]
        tab.setdefault(short, {})[c] = (n, tot / max(n, 1))
        if c not in counters:
            counters.append(c)
    counters.sort()
    with open(out, 'w') as f:
        f.write('# SQ counters per kernel (average per dispatch), %s\n\n' % db.split('/')[-1])
        f.write('| kernel | dispatches | ' + ' | '.join(counters) + ' | MFMA pipe busy |\n')
        f.write('|---|---:|' + '---:|' * (len(counters) + 1) + '\n')
        for k in sorted(tab, key=lambda k: -tab[k].get(counters[0], (0, 0))[1]):
            v = tab[k]
            n = max(x[0] for x in v.values())
            busy = ''
            if 'SQ_VALU_MFMA_BUSY_CYCLES' in v and 'SQ_BUSY_CU_CYCLES' in v and v['SQ_BUSY_CU_CYCLES'][1] > 0:
                busy = '%.3f' % (v['SQ_VALU_MFMA_BUSY_CYCLES'][1] / (4.0 * v['SQ_BUSY_CU_CYCLES'][1]))
            f.write('| `%s` | %d | ' % (k[:70], n) + ' | '.join('%.4g' % v[c][1] if c in v else '' for c in counters) +
                    ' | %s |\n' % busy)
    print(open(out).read())


if __name__ == '__main__':
    main()
