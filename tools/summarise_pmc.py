"""Summarise a rocprofv3 --pmc csv directory: per kernel name, dispatch count and mean / sum of every counter."""
import csv, glob, sys
from collections import defaultdict

rows = []
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    with open(f) as fh:
        rows += list(csv.DictReader(fh))
if not rows:
    print('no counter_collection.csv found under', sys.argv[1])
    sys.exit(0)
acc = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for r in rows:
    k = r.get('Kernel_Name', '?').split('(')[0][-60:]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    disp[k].add(r.get('Dispatch_Id'))
for k in sorted(acc, key=lambda k: -len(disp[k])):
    n = len(disp[k])
    print(f'{k}  dispatches={n}')
    for c, v in sorted(acc[k].items()):
        print(f'    {c:28s} sum={v:.6g}  per_dispatch={v / n:.6g}')
