#!/usr/bin/env python
"""Summarises a rocprofv3 --pmc run (sqlite .db): per kernel name, average counter value per dispatch.
   python tools/pmc_summary.py <results.db> [name-substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
sub = sys.argv[2] if len(sys.argv) > 2 else ''
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
view = 'counters_collection' if 'counters_collection' in tabs else None
if view is None:
    print('no counters_collection view; tables:', [t for t in tabs if 'pmc' in t.lower() or 'counter' in t.lower()])
    sys.exit(0)
cols = [d[0] for d in cur.execute('select * from %s limit 1' % view).description]
rows = cur.execute('select kernel_name, counter_name, count(*), avg(value), sum(value) from %s group by kernel_name, counter_name' % view).fetchall()
for k, c, n, avg, tot in rows:
    if sub in k:
        print('%-70s %-14s dispatches=%5d avg=%14.1f' % (k[:70], c, n, avg))
