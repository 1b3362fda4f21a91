"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel-name mean of every counter over its dispatches.
usage: pmc_summary.py DIR [name-substring]"""
import csv, glob, os, sys
from collections import defaultdict

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ''
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(dict)
for path in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
    with open(path, newline='') as f:
        for r in csv.DictReader(f):
            name = r['Kernel_Name']
            if want and want not in name:
                continue
            short = name.split('(')[0][-60:]
            acc[short][r['Counter_Name']].append(float(r['Counter_Value']))
            dur[short][r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for k, cs in acc.items():
    d = list(dur[k].values())
    print('%s: %d dispatches, %.1f us mean' % (k, len(d), sum(d) / len(d)))
    for c, v in sorted(cs.items()):
        print('   %-32s %16.0f' % (c, sum(v) / len(v)))
