"""One record per launch of one forward, joined from four runs of the same bench command on the same tuned plan:
  * the rocprofv3 kernel trace (duration of every launch, averaged over the traced steps by position),
  * the per-op list bench.py --ops-json wrote (label, FLOPs and minimum bytes of every plan op, in launch order),
  * the FETCH_SIZE and the WRITE_SIZE counter passes (HBM bytes per launch; FETCH_SIZE counts the 128-byte requests of
    wide coalesced reads as 64 bytes on gfx950 and is doubled),
  * the SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE pass (MFMA pipe occupancy per launch).
A forward is the run of launches from one launch of the plan's first kernel to the next; NMS kernels and memsets (the
other stream) are dropped by name, so position i is op i of the plan.
usage: roofline_table.py OPS.json TRACE.csv FETCH_DIR WRITE_DIR MFMA_DIR OUT.json [key=value ...]"""
import csv, glob, json, os, re, sys

FIRST = ('focus_conv_kernel', 'stem2_kernel', 'stem_conv_kernel', 'pack_image', 'normalize')
PEAK_TFLOPS, PEAK_GBS = 2500.0, 8000.0


def short(name):
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    return re.sub(r'\(.*$', '', name)[:90]


def forwards(rows, n_ops):
    """rows: (order key, kernel name, value) -> list of forwards, each a list of n_ops (name, value)."""
    rows = [r for r in sorted(rows) if 'nms_' not in r[1] and 'fillBuffer' not in r[1] and 'scale_coords' not in r[1]]
    first = rows and next((r[1] for r in rows if any(k in r[1] for k in FIRST)), None)
    if not first:
        raise SystemExit('no launch of a first kernel found')
    marks = [i for i, r in enumerate(rows) if r[1] == first]
    out = [rows[a:b] for a, b in zip(marks, marks[1:] + [len(rows)])]
    return [[(r[1], r[2]) for r in f] for f in out if len(f) == n_ops]


def counters(root):
    per = {}
    for path in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
        with open(path, newline='') as f:
            for r in csv.DictReader(f):
                d = per.setdefault(int(r['Dispatch_Id']), {'name': r['Kernel_Name']})
                d[r['Counter_Name']] = d.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    return per


def mean_by_position(fs, n_ops):
    fs = fs[-20:]
    if not fs:
        return None
    return [(fs[0][i][0], sum(f[i][1] for f in fs) / len(fs)) for i in range(n_ops)]


def main():
    ops_path, trace, fetch_dir, write_dir, mfma_dir, out_path = sys.argv[1:7]
    extra = dict(kv.split('=', 1) for kv in sys.argv[7:])
    ops = json.load(open(ops_path))
    n = len(ops['ops'])
    rows = []
    with open(trace, newline='') as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), r['Kernel_Name'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
    dur = mean_by_position(forwards(rows, n), n)
    if dur is None:
        raise SystemExit('no traced forward has %d launches' % n)
    fetch = counters(fetch_dir)
    write = counters(write_dir)
    mfma = counters(mfma_dir)
    rd = mean_by_position(forwards([(d, v['name'], v.get('FETCH_SIZE', 0.0) * 1024 * 2) for d, v in fetch.items()], n), n)
    wr = mean_by_position(forwards([(d, v['name'], v.get('WRITE_SIZE', 0.0) * 1024) for d, v in write.items()], n), n)
    mb = mean_by_position(forwards([(d, v['name'], v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)) for d, v in mfma.items()], n), n)
    ga = mean_by_position(forwards([(d, v['name'], v.get('GRBM_GUI_ACTIVE', 0.0)) for d, v in mfma.items()], n), n)
    table = []
    for i, op in enumerate(ops['ops']):
        us = dur[i][1]
        row = {'launch': i, 'kernel': short(dur[i][0]), 'label': op['label'], 'kind': op['kind'], 'us': round(us, 2),
               'tflops': round(op['flops'] / us / 1e6, 1), 'plan_bytes': op['bytes'], 'plan_tb_s': round(op['bytes'] / us / 1e6, 3)}
        if rd and wr:
            row['counter_bytes'] = round(rd[i][1] + wr[i][1])
            row['counter_tb_s'] = round((rd[i][1] + wr[i][1]) / us / 1e6, 3)
        if mb and ga and ga[i][1] > 0:
            row['mfma_busy'] = round(mb[i][1] / (1024.0 * ga[i][1] / 8.0), 4)
        table.append(row)
    conv = [r for r in table if r['kind'] in ('conv', 'focus_conv', 'detect')]
    tot_us = sum(r['us'] for r in conv)
    flops = sum(op['flops'] for op in ops['ops'] if op['kind'] in ('conv', 'focus_conv', 'detect'))
    out = dict(extra)
    out.update({'model': ops['model'], 'batch': ops['batch'], 'dtype': ops['dtype'], 'launches_per_forward': n,
                'sources': {'durations': 'rocprofv3 --kernel-trace of bench.py, mean over the last traced steps by launch position',
                            'flops_and_plan_bytes': 'bench.py --ops-json (what each recorded launch needs at minimum)',
                            'counter_bytes': 'rocprofv3 --pmc FETCH_SIZE (x2) and WRITE_SIZE, one pass each, same plan',
                            'mfma_busy': 'SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)'},
                'convolution_launches': {'count': len(conv), 'us': round(tot_us, 1), 'tflops': round(flops / tot_us / 1e6, 1),
                                         'mfma_frac': round(flops / tot_us / 1e6 / PEAK_TFLOPS, 4),
                                         'plan_tb_s': round(sum(r['plan_bytes'] for r in conv) / tot_us / 1e6, 3),
                                         'counter_bytes': sum(r.get('counter_bytes', 0) for r in conv) or None,
                                         'counter_tb_s': round(sum(r.get('counter_bytes', 0) for r in conv) / tot_us / 1e6, 3) if rd and wr else None,
                                         'mfma_busy': round(sum(mb[r['launch']][1] for r in conv) / (1024.0 * sum(ga[r['launch']][1] for r in conv) / 8.0), 4) if mb and ga else None},
                'forward_us_sum_of_launches': round(sum(r['us'] for r in table), 1),
                'top_kernels': sorted(table, key=lambda r: -r['us'])[:12], 'all_launches': table})
    json.dump(out, open(out_path, 'w'), indent=1)
    print('%s batch %d: %d launches, %.1f us; convolution launches %.1f us, %.1f TFLOP/s' % (ops['model'], ops['batch'], n, out['forward_us_sum_of_launches'], tot_us, flops / tot_us / 1e6))
    for r in out['top_kernels']:
        print('%3d %8.1f us %7.1f TF/s %6.2f TB/s plan %6s TB/s counters  MFMA %5s  %s' % (
            r['launch'], r['us'], r['tflops'], r['plan_tb_s'], r.get('counter_tb_s', '-'), r.get('mfma_busy', '-'), r['label'][:90]))


if __name__ == '__main__':
    main()
