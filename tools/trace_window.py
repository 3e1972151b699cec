"""Every kernel of a window of a rocprofv3 --kernel-trace CSV in start order: offset from the window's first start, duration, queue.
usage: python tools/trace_window.py <kernel_trace.csv> <anchor substring> <first anchor launch> <anchor launches>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
anchor, first, cnt = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
idx = [i for i, r in enumerate(rows) if anchor in r['Kernel_Name']]
a, b = idx[first], idx[first + cnt]
t0 = rows[a]['s']
for r in rows[a:b + 1]:
    print('%9.1f us  +%7.1f us  q%-3s %s' % ((r['s'] - t0) / 1e3, (r['e'] - r['s']) / 1e3, r.get('Queue_Id', '?'), r['Kernel_Name'][:90]))
