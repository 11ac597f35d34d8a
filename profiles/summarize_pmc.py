#!/usr/bin/env python3
"""Per-kernel average of one rocprofv3 --pmc counter from a rocpd sqlite database.
usage: summarize_pmc.py results.db COUNTER   -> prints kernel, dispatches, avg, max (counter units)"""
import sqlite3
import sys


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    ksym = [r[1] for r in c.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "display_name" if "display_name" in ksym else "kernel_name"
    q = ("select s.%s, count(*), avg(v), max(v), sum(v) from (select d.kernel_id as kid, sum(e.value) as v "
         "from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id "
         "join rocpd_kernel_dispatch d on d.event_id = e.event_id where p.name = ? group by d.id) x "
         "join rocpd_info_kernel_symbol s on s.id = x.kid group by s.%s order by 5 desc" % (name_col, name_col))
    return list(c.execute(q, (counter,)))


if __name__ == "__main__":
    for r in per_kernel(sys.argv[1], sys.argv[2]):
        print("%-70s n=%4d avg=%14.1f max=%14.1f" % (r[0][:70], r[1], r[2], r[3]))
