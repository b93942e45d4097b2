"""Aggregate a rocprofv3 counter_collection.csv per kernel name: sum of each counter / dispatch count."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
for r in rows:
    k = r.get('Kernel_Name', '')[:60]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[k].add(r.get('Dispatch_Id'))
for k in sorted(agg, key=lambda k: -sum(agg[k].values())):
    n = max(len(cnt[k]), 1)
    print(k, 'dispatches', n, ' '.join('%s=%.4g' % (c, v / n) for c, v in sorted(agg[k].items())))
