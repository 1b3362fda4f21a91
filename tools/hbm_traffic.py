"""HBM traffic of one bench step from two rocprofv3 --pmc passes over bench.py (FETCH_SIZE and WRITE_SIZE, one pass
each).  One forward = the dispatches between the last two launches of the plan's first kernel (the Focus kernel); one
NMS call = those between the last two launches of its first kernel.  FETCH_SIZE is doubled: on gfx950 it tallies the
128-byte requests of wide coalesced reads at 64 bytes (narrower loads, as in the Focus kernel, are then over-counted).
usage: hbm_traffic.py FETCH_DIR WRITE_DIR OUT.json [MODEL BATCH]"""
import csv, glob, json, os, sys
from collections import defaultdict


def one_step(root, counter):
    rows = []
    for path in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
        with open(path, newline='') as f:
            for r in csv.DictReader(f):
                if r['Counter_Name'] == counter:
                    rows.append((int(r['Dispatch_Id']), r['Kernel_Name'], float(r['Counter_Value'])))
    rows.sort()
    out = []
    for first, keep in ((('focus_conv_kernel', 'stem2_kernel', 'stem_conv_kernel'), lambda n: 'nms_' not in n), (('nms_candidates_kernel',), lambda n: 'nms_' in n)):
        marks = [i for i, r in enumerate(rows) if any(k in r[1] for k in first)]
        if len(marks) < 2:
            raise SystemExit('fewer than two launches of %s in %s' % ('/'.join(first), root))
        out += [r for r in rows[marks[-2]:marks[-1]] if keep(r[1])]
    return out


def family(name):
    for key, fam in (('csp1_kernel', 'conv_csp'), ('detect_p_kernel', 'conv_detect'), ('stem2_kernel', 'conv_stem2')):
        if key in name:
            return fam
    for key in ('focus_conv', 'conv_ph', 'conv_lw', 'conv_lin3', 'conv_c64', 'conv_c128', 'conv_p1x1', 'conv_s2c64', 'conv_s2c32', 'conv_igemm', 'conv_halo', 'conv_ws1x1', 'conv3x3_ws', 'detect_decode', 'nms_', 'maxpool', 'upsample', 'copy_channels'):
        if key in name:
            return key
    return 'other'


fetch = one_step(sys.argv[1], 'FETCH_SIZE')
write = one_step(sys.argv[2], 'WRITE_SIZE')
per = defaultdict(lambda: [0.0, 0.0, 0])
for _, name, v in fetch:
    per[family(name)][0] += v * 1024 * 2   # KiB -> bytes, x2 (see the docstring)
    per[family(name)][2] += 1
for _, name, v in write:
    per[family(name)][1] += v * 1024
model = sys.argv[4] if len(sys.argv) > 4 else 'yolov5s'
batch = int(sys.argv[5]) if len(sys.argv) > 5 else 32
out = {'model': model, 'batch': batch, 'unit': 'bytes per bench step', 'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE x2',
       'kernels': {k: {'read': r, 'write': w, 'launches': n} for k, (r, w, n) in sorted(per.items())}}
conv = [v for k, v in per.items() if 'conv' in k]
out['conv_total'] = sum(v[0] + v[1] for v in conv)
out['step_total'] = sum(v[0] + v[1] for v in per.values())
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps(out, indent=1))
