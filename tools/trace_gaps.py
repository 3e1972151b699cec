"""Timeline of a rocprofv3 --kernel-trace CSV: for the dominant kernel (by total time) the launch-to-launch period, its duration and the
gap before each launch; what ran inside the gaps.  usage: python tools/trace_gaps.py <kernel_trace.csv> [first] [count]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 40
count = int(sys.argv[3]) if len(sys.argv) > 3 else 6
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
tot = defaultdict(int)
for r in rows:
    tot[r['Kernel_Name']] += r['e'] - r['s']
dom = max(tot, key=tot.get)
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'] == dom]
print('dominant:', dom[:80], 'launches', len(idx))
for a, b in zip(idx[first:first + count], idx[first + 1:first + count + 1]):
    ra, rb = rows[a], rows[b]
    print('sum %.3f ms | end->next start %.3f ms | period %.3f ms' % ((ra['e'] - ra['s']) / 1e6, (rb['s'] - ra['e']) / 1e6, (rb['s'] - ra['s']) / 1e6))
    for r in rows[a + 1:b]:
        print('     +%.3f  %.3f ms  q%s  %s' % ((r['s'] - ra['s']) / 1e6, (r['e'] - r['s']) / 1e6, r.get('Queue_Id', '?'), r['Kernel_Name'][:70]))
if len(sys.argv) > 4:
    # periods of the dominant kernel over a range of launches
    lo, hi = int(sys.argv[4]), int(sys.argv[5])
    per = [(rows[idx[i + 1]]['s'] - rows[idx[i]]['s']) / 1e6 for i in range(lo, hi)]
    print('periods [%d, %d):' % (lo, hi), ' '.join('%.2f' % p for p in per))
    print('span %.3f ms over %d launches = %.3f ms each' % ((rows[idx[hi]]['e'] - rows[idx[lo]]['s']) / 1e6, hi - lo + 1, (rows[idx[hi]]['e'] - rows[idx[lo]]['s']) / 1e6 / (hi - lo + 1)))
