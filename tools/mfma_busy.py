"""MFMA pipe occupancy per launch of one forward from a rocprofv3 --pmc pass over bench.py with
SQ_VALU_MFMA_BUSY_CYCLES (cycles an MFMA pipe is busy, summed over the SIMDs) and GRBM_GUI_ACTIVE (cycles the dispatch
kept the GPU busy, summed over the 8 XCDs): busy fraction = MFMA cycles / (1024 SIMDs x GUI_ACTIVE / 8).  One forward =
the dispatches between the last two launches of the plan's first kernel.  usage: mfma_busy.py PMC_DIR"""
import csv, glob, os, re, sys
from collections import defaultdict
vals = defaultdict(dict)
names = {}
for path in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    with open(path, newline='') as f:
        for r in csv.DictReader(f):
            d = int(r['Dispatch_Id'])
            vals[d][r['Counter_Name']] = vals[d].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
            names[d] = r['Kernel_Name']
ids = sorted(vals)
marks = [i for i, d in enumerate(ids) if 'focus_conv_kernel' in names[d] or 'stem2_kernel' in names[d]]
if len(marks) < 2:
    raise SystemExit('fewer than two launches of the first kernel')
def short(name):
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    return re.sub(r'\(.*$', '', name)[:100]
tm = tg = 0.0
print('launch  MFMA-busy  kernel')
for k, d in enumerate(ids[marks[-2]:marks[-1]]):
    if 'nms_' in names[d] or 'fillBuffer' in names[d]:
        continue
    m, g = vals[d].get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0), vals[d].get('GRBM_GUI_ACTIVE', 0.0)
    tm += m
    tg += g
    print('%3d    %6.1f %%   %s' % (k, 100.0 * m / (1024.0 * g / 8.0) if g else 0.0, short(names[d])))
print('whole forward: %.1f %% (MFMA cycles %.3g over 1024 SIMDs x %.3g cycles)' % (100.0 * tm / (1024.0 * tg / 8.0), tm, tg / 8.0))
