import collections, re, sys
agg = collections.defaultdict(list)
for ln in open(sys.argv[1]):
    m = re.search(r'\[prisim_hip host\] (.*?): ([0-9.]+) us', ln)
    if m: agg[m.group(1)].append(float(m.group(2)))
    m = re.search(r'\[prisim_hip alloc\] (\d+) B \(had (\d+)\): ([0-9.]+) us', ln)
    if m: agg['ALLOC'].append(float(m.group(3)))
for k, v in sorted(agg.items()):
    v.sort()
    print('%-70s n=%5d sum %.1f ms  median %.1f  p99 %.1f  max %.1f us' % (k[:70], len(v), sum(v) / 1e3, v[len(v) // 2], v[int(len(v) * 0.99)], v[-1]))
