"""Merged timeline of HIP API calls (host) and kernels (device) from rocprofv3 --hip-trace --kernel-trace CSVs, for a window anchored at
the n-th launch of a kernel.  usage: python tools/trace_merge.py <dir/prefix> <anchor kernel substring> <first> <count>"""
import csv, glob, sys
prefix, anchor, first, cnt = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
kern = list(csv.DictReader(open(glob.glob(prefix + '*kernel_trace.csv')[0])))
api = list(csv.DictReader(open(glob.glob(prefix + '*hip_api_trace.csv')[0])))
ev = []
for r in kern:
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'GPU q%s' % r.get('Queue_Id', '?'), r['Kernel_Name'][:60], r.get('Correlation_Id', '')))
for r in api:
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'host', r['Function'], r.get('Correlation_Id', '')))
ev.sort()
idx = [i for i, e in enumerate(ev) if e[2].startswith('GPU') and anchor in e[3]]
a, b = idx[first], idx[first + cnt]
t0 = ev[a][0]
# start a little before the anchor to catch the launch call
lo = a
while lo > 0 and ev[lo][0] > t0 - 120000:
    lo -= 1
for e in ev[lo:b + 1]:
    print('%9.1f us  +%7.1f us  %-8s %-62s %s' % ((e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[2], e[3], e[4]))
