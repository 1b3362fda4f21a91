"""Where the host's time per step goes (wall clock of the calls, no synchronisation inside the loops):
python tools/host_floor.py [--model yolov5s --batch 32]"""
import argparse
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def wall(fn, n=200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    return dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='yolov5s')
    ap.add_argument('--batch', type=int, default=32)
    args = ap.parse_args()
    import bench
    from yolo_amd import engine, nms as hip_nms, _lib
    torch.set_grad_enabled(False)
    dev = torch.device('cuda:0')
    model, _, _ = bench.build_model(args.model, dev)
    x = torch.rand(args.batch, 3, 640, 640).half().to(dev)
    for _ in range(3):
        z, _ = model(x)
    eng = engine.engine_of(model)
    plan = next(iter(eng.plans.values()))
    side = torch.cuda.Stream(dev)
    ext = plan._ext(x, z, [torch.empty(s, dtype=plan.dtype, device=dev) for s in plan.x_shapes])
    stream = _lib.current_stream()
    n = 30 if args.batch >= 16 else 300
    rows = [
        ('engine.state() (cached tensor list + hook tables)', wall(eng.state, n)),
        ('_plan_if_eligible', wall(lambda: engine._plan_if_eligible(model, x), n)),
        ('y5_plan_run alone (%d launches)' % plan.lib.y5_plan_num_ops(plan.handle), wall(lambda: plan.lib.y5_plan_run(plan.handle, ext, len(ext), stream), n)),
        ('plan.run (outputs + y5_plan_run)', wall(lambda: plan.run(x), n)),
        ('model(x)', wall(lambda: model(x), n)),
        ('nms_padded (same stream)', wall(lambda: hip_nms.nms_padded(z, 0.001, 0.65, check_overflow=False), n)),
        ('nms_padded (side stream)', wall(lambda: hip_nms.nms_padded(z, 0.001, 0.65, check_overflow=False, stream=side), n)),
    ]
    engine.set_output_ring(model, 2)
    post = hip_nms.PaddedNms(z.shape[0], z.shape[1], z.shape[2], z.dtype, dev, conf_thres=0.001, iou_thres=0.65, ring=2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def step():
        post.wait_oldest()
        e0.record()
        zz, _ = model(x)
        e1.record()
        post(zz, stream=side)
    rows += [
        ('model(x), outputs from a ring of two', wall(lambda: model(x), n)),
        ('PaddedNms (same stream)', wall(lambda: post(z), n)),
        ('PaddedNms (side stream)', wall(lambda: post(z, stream=side), n)),
        ('two event records', wall(lambda: (e0.record(), e1.record()), n)),
        ('whole step as bench.py issues it', wall(step, n)),
    ]
    for name, ms in rows:
        print('%-55s %.4f ms' % (name, ms))


if __name__ == '__main__':
    main()
