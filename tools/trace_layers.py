"""Per-launch table of one forward from a rocprofv3 kernel trace of bench.py: the launches between consecutive first
kernels of the plan (Focus, or the fused stem), averaged over the traced steps by position, so that row i is launch i of the plan (tools/profile_ops.py prints
the same rows from hipEvent pairs).  usage: trace_layers.py kernel_trace.csv"""
import csv, re, sys
rows = []
with open(sys.argv[1], newline='') as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
marks = [i for i, r in enumerate(rows) if 'focus_conv_kernel' in r[2] or 'stem2_kernel' in r[2] or 'stem_conv_kernel' in r[2]]
steps = [rows[a:b] for a, b in zip(marks[:-1], marks[1:])]
n = max(set(len(s) for s in steps), key=[len(s) for s in steps].count)   # the steady-state launch count
steps = [s for s in steps if len(s) == n][-20:]
def short(name):
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    return re.sub(r'\(.*$', '', name)[:110]
print('%d steps of %d launches (forward + NMS)' % (len(steps), n))
tot = conv = 0.0
for i in range(n):
    d = sum(s[i][1] - s[i][0] for s in steps) / len(steps) / 1e3
    name = short(steps[0][i][2])
    tot += d
    if 'conv' in name or 'csp1' in name or 'detect_p' in name or 'stem2' in name:
        conv += d
    print('%3d %8.1f us  %s' % (i, d, name))
span = sum(s[-1][1] - s[0][0] for s in steps) / len(steps) / 1e3
print('sum of kernel durations per step %.1f us (convolution kernels %.1f us); first start -> last end %.1f us' % (tot, conv, span))
