#!/usr/bin/env python3
"""Kernel totals out of the SQLite database rocprofv3 writes by default (`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- prog`
leaves DIR/NAME_results.db): name, calls, total ms, average ms, largest first.
    python benchmarks/rocpd_top.py gpurun_out/prof/NAME_results.db"""
import sqlite3,sys
c=sqlite3.connect(sys.argv[1])
for r in c.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc limit 8"): print(r[0][:70].replace("(anonymous namespace)::",""), r[1], round(r[2]/1e6,2), round(r[3]/1e6,3))
