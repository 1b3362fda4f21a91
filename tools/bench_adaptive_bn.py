"""Times the BatchNorm-recalibration forward (Training.adaptive_BN, filter_pruning/fp_train.py:231-240) on one GPU:
the HIP path (yolo_amd.adaptive_bn.Recalibrator), and beside it the same train-mode forward through the torch modules in
fp32 on the same GPU (what the reference executes) and in fp16.  Prints one JSON line; --ops adds per-kernel times.

    python tools/bench_adaptive_bn.py [--model yolov5s] [--batch 32] [--size 640] [--steps 10] [--ops]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import util  # noqa: E402


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='yolov5s')
    ap.add_argument('--pruned', action='store_true')
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--size', type=int, default=640)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--ops', action='store_true')
    ap.add_argument('--no-torch', action='store_true')
    a = ap.parse_args()
    from yolo_amd.adaptive_bn import Recalibrator
    model = util.build_model(a.model)
    if a.pruned:
        util.prune_like(model)
    model = model.cuda().train()
    x = torch.randint(0, 256, (a.batch, 3, a.size, a.size), dtype=torch.uint8, device='cuda')
    rec = Recalibrator(model)
    rec.step(x)   # builds and tunes the program
    ms = timed(lambda: rec.step(x), a.steps, a.warmup)
    rec.finish()
    prog = next(iter(rec._programs.values()))
    out = {'metric': 'adaptive_bn_images_per_sec', 'value': round(a.batch / ms * 1e3, 1), 'unit': 'images/s', 'ms_per_batch': round(ms, 3),
           'config': {'workload': '%s%s train-mode forward, batch %d, %dx%d uint8 input' % (a.model, ' pruned' if a.pruned else '', a.batch, a.size, a.size)},
           'dtype': 'f16 activations, f32/f64 statistics', 'kernels': len(prog.ops)}
    if not a.no_torch:
        with torch.no_grad():
            ms32 = timed(lambda: model(x.float() / 255.0), max(2, a.steps // 3), 2)
            half = __import__('copy').deepcopy(model).half()
            ms16 = timed(lambda: half(x.half() / 255.0), max(2, a.steps // 3), 2)
        out['torch_fp32_same_gpu'] = {'ms_per_batch': round(ms32, 3), 'images_per_sec': round(a.batch / ms32 * 1e3, 1)}
        out['torch_fp16_same_gpu'] = {'ms_per_batch': round(ms16, 3), 'images_per_sec': round(a.batch / ms16 * 1e3, 1)}
    print(json.dumps(out))
    if a.ops:
        from yolo_amd import _lib
        stream = _lib.current_stream()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(len(prog.ops) + 1)]
        tot = {}
        for rep in range(3):
            e[0].record()
            for i, (fn, args) in enumerate(prog.ops):
                fn(*args, stream)
                e[i + 1].record()
            torch.cuda.synchronize()
            if rep == 2:
                for i, label in enumerate(prog.labels):
                    us = e[i].elapsed_time(e[i + 1]) * 1e3
                    kind = 'bn stats' if label.endswith('bn stats') else 'bn finalize' if label.endswith('finalize') else \
                        'bn apply' if label.endswith('bn apply') else 'conv' if ' conv' in label else 'other'
                    tot[kind] = tot.get(kind, 0.0) + us
                    print('%3d %9.1f us  %s' % (i, us, label))
        print('by kind (us):', {k: round(v, 1) for k, v in tot.items()}, 'sum', round(sum(tot.values()), 1))
        # bytes the BatchNorm kernels must move: stats read every normalised tensor once, apply reads and writes it
        # (and reads the shortcut where there is one)
        nbytes = {'bn stats': 0, 'bn apply': 0}
        for (fn, args), label in zip(prog.ops, prog.labels):
            if label.endswith('bn stats'):
                nbytes['bn stats'] += args[2] * args[3] * 2
            elif label.endswith('bn apply'):
                nbytes['bn apply'] += args[6] * args[7] * 2 * (3 if args[4] else 2)
        for k, b in nbytes.items():
            print('%s: %.3f GB algorithmic, %.0f GB/s' % (k, b / 1e9, b / 1e3 / tot[k]))


if __name__ == '__main__':
    main()
