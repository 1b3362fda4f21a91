"""Benchmark of the hot path: yolov5s 640x640 fp16, conv forward + Detect decode + NMS, images/s.

  python bench.py [--gpus N --steps K --warmup W]
      N = 1: one process.  N > 1 from a plain call: this process starts N rank processes (one per GPU, before it has
      touched a GPU itself) and passes rank 0's JSON line on; under torch.distributed.run (WORLD_SIZE set) it IS a rank.
  python bench.py --gpus N --model yolov5x --global-batch 256
      the sharded form: the global batch is split into contiguous slices (128 / 64 / 32 images per GPU at N = 2 / 4 / 8),
      "scaling": "strong".  Without --global-batch every GPU gets --batch images (weak scaling).

One step = one batch of synthetic images (already resident in HBM as fp16 NCHW) through the HIP launch
plan (all convs, SPP, upsample/concat, head convs, decode) and the batched NMS kernels, ending with padded
detections + counts on the device; with N>1 the step ends with the all_gather of the detection blocks (RCCL; gloo
when rehearsing more ranks than there are GPUs).  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

PEAK_MFMA_TFLOPS = 2500.0   # dense fp16 MFMA peak of one MI355X
PEAK_HBM_GBS = 8000.0       # HBM3E peak of one MI355X
DECODE_BYTES_PER_IMAGE_640 = 2 * 25200 * 85 * 2.0   # Detect decode at 640x640, nc = 80: head tensor read + z written, fp16


def reference_layer_work(model, x):
    """(bytes, FLOPs) of one forward of the batch x with every nn.Conv2d of the graph as its own kernel: fp16 input +
    output per image, weights once, 2 FLOP per multiply-add (Detect's convolutions included; BatchNorm folded,
    activations free).  Shapes come from forward hooks on the torch modules for ONE image (the hooks keep the HIP engine
    out of that forward), outside every timed region."""
    import torch.nn as nn
    recs = []
    hooks = [m.register_forward_hook(lambda m, i, o: recs.append((m, tuple(i[0].shape), tuple(o.shape))))
             for m in model.modules() if isinstance(m, nn.Conv2d)]
    try:
        with torch.no_grad():
            one = x[:1]
            model(one.half() / 255.0 if one.dtype is torch.uint8 else one)
    finally:
        for h in hooks:
            h.remove()
    act = wts = flops = 0
    for m, i, o in recs:
        act += 2 * (i[1] * i[2] * i[3] + o[1] * o[2] * o[3])
        wts += 2 * m.weight.numel()
        flops += 2 * o[1] * o[2] * o[3] * m.weight[0].numel()
    b = x.shape[0]
    return float(act * b + wts), float(flops * b)


def build_model(name, device, head_gain=2.0, dtype=torch.float16):
    import util  # tests/util.py: seeded construction + BatchNorm calibration + head gain (synthetic weights)
    model = util.build_model(name, seed=0)
    util.calibrate(model, seed=1, size=640, head_gain=head_gain)
    fp32_fused = util.quiet(model.fuse).eval()
    import copy
    cpu_sd = {k: v.clone() for k, v in fp32_fused.state_dict().items()}
    gpu_model = copy.deepcopy(fp32_fused).to(dtype).to(device).eval()
    return gpu_model, cpu_sd, float(max(fp32_fused.stride))


def _cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.lower().startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(name, cpu_sd, strides, conf, iou, budget_s=14.0):
    """The CPU oracle (torch-CPU fp32 restatement of the reference + C NMS) on a bounded sample of the same workload:
    a few single images (batch 1, the reference's CPU-runnable case) and batches of 32 (the GPU workload's batch)."""
    import yaml
    from oracle import y5ref, nms_c
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    cfg = yaml.safe_load(open(os.path.join(ROOT, 'yolo_amd', 'models', name + '.yaml')))
    g = torch.Generator().manual_seed(1)

    def run(batch, budget, most):
        done, t_total = 0, 0.0
        with torch.no_grad():
            while t_total < budget and done < most:
                x = torch.rand(batch, 3, 640, 640, generator=g).half().float()
                t0 = time.time()
                z, _ = y5ref.forward(cfg, cpu_sd, x, precision='fp32', strides=strides)
                nms_c.non_max_suppression(z.numpy(), conf, iou)
                t_total += time.time() - t0
                done += batch
        return done, t_total

    run(1, 0.0, 1)  # first call pays thread-pool start-up and lazy initialisation
    n1, t1 = run(1, budget_s * 0.25, 8)
    n32, t32 = run(32, budget_s * 0.75, 96)
    return {'value': round(n32 / t32, 2), 'unit': 'images/s', 'cores': threads, 'kind': 'port', 'cpu': _cpu_model(),
            'batch1_images_per_s': round(n1 / t1, 2),
            'sample': '%d images in batches of 32 (value) and %d single images (batch1_images_per_s) of the same synthetic '
                      '640x640 workload: fp32 torch-CPU forward+decode, C NMS' % (n32, n1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--model', default='yolov5s')
    ap.add_argument('--batch', type=int, default=32, help='images per GPU and step (weak scaling)')
    ap.add_argument('--global-batch', type=int, default=0, help='images per step over all GPUs, sharded (strong scaling)')
    ap.add_argument('--conf', type=float, default=0.001)
    ap.add_argument('--iou', type=float, default=0.65)
    ap.add_argument('--head-gain', type=float, default=2.0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--serial-nms', action='store_true', help='NMS on the forward\'s own stream (no overlap with the next step)')
    ap.add_argument('--profile-passes', type=int, default=5)
    ap.add_argument('--dtype', choices=('f16', 'bf16'), default='f16', help='element type of weights and activations (accumulation is fp32 either way)')
    ap.add_argument('--no-latency', action='store_true', help='skip the batch-1 latency measurement (profiler runs: keeps other plans out of the trace)')
    ap.add_argument('--ops-json', default='', help='write the per-op list of the plan (label, kind, ms, FLOPs, minimum bytes) to this file')
    ap.add_argument('--in-flight', type=int, default=3, help='steps in flight, each on its own stream with its own launch plan (yolo_amd.serving.Pipeline); 1 = forward on one stream, NMS beside the next forward on a second')
    ap.add_argument('--graph', action='store_true', help='replay the forward as captured HIP graphs (one launch per forward): less host time per step, same device time')
    ap.add_argument('--fresh-outputs', action='store_true', help='allocate output tensors every step (the reference\'s behaviour) instead of reusing two sets')
    args = ap.parse_args()

    from yolo_amd import dist as ydist
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: become the supervisor of N rank processes; nothing here has touched a GPU
        raise SystemExit(ydist.launch_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus))
    rank, world, local = ydist.init_from_env()
    if world != args.gpus:
        raise SystemExit('bench: --gpus %d but WORLD_SIZE is %d' % (args.gpus, world))
    device = torch.device('cuda', local % max(1, torch.cuda.device_count()))  # ranks share devices only in a rehearsal
    pinned = ydist.pin_cpus(local, int(os.environ.get('LOCAL_WORLD_SIZE', world)))
    torch.cuda.set_device(device)
    torch.set_grad_enabled(False)
    if args.global_batch:
        if args.global_batch % world:
            raise SystemExit('bench: --global-batch %d is not a multiple of %d GPUs (the gather moves equal blocks)' % (args.global_batch, world))
        lo, hi = ydist.shard_bounds(args.global_batch, rank, world)
        args.batch = hi - lo
        if args.batch <= 0:
            raise SystemExit('bench: --global-batch %d leaves rank %d without images' % (args.global_batch, rank))
    global_batch = args.global_batch or args.batch * world

    from yolo_amd import engine, nms as hip_nms
    import torch.distributed as dist
    dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float16
    model, cpu_sd, _ = build_model(args.model, device, args.head_gain, dtype)
    strides = model.stride
    g = torch.Generator().manual_seed(1 + rank)
    x = torch.rand(args.batch, 3, 640, 640, generator=g).to(dtype).to(device)

    local_out = [None]

    # The NMS kernels are small and latency-bound: they run on a second stream, ordered after the forward that feeds them,
    # so that the next step's forward (issued on the main stream right behind) fills the chip beside them.  Every step
    # still computes its own forward, decode, NMS (and exchange); fence() waits for both streams.
    side = None if args.serial_nms else torch.cuda.Stream(device)

    # A serving loop allocates nothing per step: the launch plan writes into two sets of output tensors in turn and the
    # NMS into two sets of its own (made once, below); step k's results stay valid until step k + 2 is issued.
    # kernel variants are timed by rank 0 only; the other ranks replay its choices
    tuned_elsewhere = ydist.tune_once(lambda: model(x))
    post, graphed, pipe = None, None, None
    if args.in_flight > 1 and not args.fresh_outputs and not args.graph and not args.serial_nms:
        from yolo_amd import serving
        try:   # every step in flight has activation buffers of its own: a very large batch may not fit three times
            pipe = serving.Pipeline(model, x, depth=args.in_flight, ring=2, conf_thres=args.conf, iou_thres=args.iou)
            side = None
        except torch.cuda.OutOfMemoryError:
            engine.engine_of(model)._drop_lanes()
            torch.cuda.empty_cache()
            args.in_flight = 1
    if pipe is None and not args.fresh_outputs:
        engine.set_output_ring(model, 2)
        z, _ = model(x)
        if args.graph:
            graphed = engine.GraphedForward(model, x, sets=2)
        post = hip_nms.PaddedNms(z.shape[0], z.shape[1], z.shape[2], z.dtype, device, conf_thres=args.conf, iou_thres=args.iou, ring=2)

    def step(events=None):
        if pipe is not None:   # forward and NMS of this step on the step's own stream; other steps' kernels run beside them
            st = pipe(x, timing=events)
            det, count, ncand = st.det, st.count, st.n_candidates
            if world > 1:
                with torch.cuda.stream(st.stream):
                    local_out[0] = (det, count)
                    det, count = ydist.gather_detections(det, count)
            return det, count, ncand
        if post is not None and side is not None:
            post.wait_oldest()  # the forward about to be issued overwrites the prediction that NMS call read on the other stream
        if events is not None:
            events[0].record()
        z, _ = model(x) if graphed is None else graphed()
        if events is not None:
            events[1].record()
        if post is not None:
            det, count, kept, ncand = post(z, stream=side)
        else:
            det, count, kept, ncand = hip_nms.nms_padded(z, args.conf, args.iou, return_index=True, check_overflow=False, stream=side)
        if world > 1:
            with torch.cuda.stream(side):
                local_out[0] = (det, count)
                det, count = ydist.gather_detections(det, count)
        return det, count, ncand

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    eng = engine.engine_of(model)
    if eng.calls < args.warmup:
        raise SystemExit('bench: forward did not run on the HIP engine')
    # Inside the timed loop: an event pair around every forward on the stream it is launched on (the device time of the
    # forward as the timed steps ran it, NMS of the previous step beside it) and the host's wall time inside step() (what
    # the Python + launch path costs per step; the device queue is never full, so step() does not wait for the GPU).
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    host_s = 0.0
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        h0 = time.perf_counter()
        out = step(ev[i])
        host_s += time.perf_counter() - h0
    fence()
    elapsed = time.perf_counter() - t0
    loop_forward_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps
    host_enqueue_ms = host_s / args.steps * 1e3
    rank_ms = [elapsed / args.steps * 1e3]
    if world > 1:
        cdev = 'cpu' if dist.get_backend() == 'gloo' else device
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        rank_ms = [float(e.item()) / args.steps * 1e3 for e in every]
        elapsed = max(float(e.item()) for e in every)   # MAX over ranks
    det, count, ncand = out
    if det.shape[0] != global_batch:
        raise SystemExit('bench: gathered %d images, expected %d' % (det.shape[0], global_batch))
    gather_ms = None
    if world > 1:  # the exchange step on its own: every rank takes part, rank 0 reports
        reps = 20
        fence()
        t0 = time.perf_counter()
        for _ in range(reps):
            ydist.gather_detections(*local_out[0])
        fence()
        t = torch.tensor([(time.perf_counter() - t0) / reps * 1e3], dtype=torch.float64, device='cpu' if dist.get_backend() == 'gloo' else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        gather_ms = float(t.item())
    cand = ncand.float().mean().item()
    kept = count.float().mean().item()
    if int(ncand.max().item()) > hip_nms.DEFAULT_CANDIDATES:
        raise SystemExit('bench: candidate capacity overflow, result invalid')

    record = None
    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = global_batch * args.steps / elapsed
        # ---- per-kernel device time from hipEvent pairs around every launch of the plan (same stream)
        acc = None
        for _ in range(args.profile_passes):
            rows = eng.profile(x)
            if acc is None:
                acc = rows
            else:
                for a, r in zip(acc, rows):
                    a['ms'] += r['ms']
        for a in acc:
            a['ms'] /= args.profile_passes
        # An event record between two launches costs the stream a bubble of its own (~2 us): the per-op durations sum
        # to more than the uninstrumented forward takes.  Time that forward with one event pair and scale the per-op
        # durations so that they sum to it.
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f0.record()
        for _ in range(args.profile_passes):
            model(x)
        f1.record()
        torch.cuda.synchronize()
        forward_ms = f0.elapsed_time(f1) / args.profile_passes
        # the per-op durations are scaled to the forward as the TIMED steps ran it (event pairs inside the loop), so the
        # roofline's denominator is a share of ms_per_step and cannot exceed it
        # (with several steps in flight the in-loop pairs measure a forward that shares the device with other steps: the
        # per-op durations are then scaled to the forward ALONE on the device, timed just above)
        event_scale = (loop_forward_ms if pipe is None else forward_ms) / sum(a['ms'] for a in acc)
        for a in acc:
            a['ms'] *= event_scale
        if args.ops_json:
            with open(args.ops_json, 'w') as f:
                json.dump({'model': args.model, 'batch': args.batch, 'dtype': args.dtype, 'ops': acc}, f, indent=1)
        det = [a for a in acc if a['kind'] == 'detect']  # Detect levels as one kernel each: head conv with the decode as its epilogue
        conv = [a for a in acc if a['kind'] in ('conv', 'focus_conv')] + det
        conv_ms = sum(a['ms'] for a in conv)
        conv_flops = sum(a['flops'] for a in conv)
        dec = [a for a in acc if a['kind'] == 'decode'] or det
        dec_ms = sum(a['ms'] for a in dec)
        dec_bytes = sum(a['bytes'] for a in dec)
        plan_ms = sum(a['ms'] for a in acc)
        # NMS stage timed as a whole with events on the same stream
        z, _ = model(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.profile_passes):
            hip_nms.nms_padded(z, args.conf, args.iou, check_overflow=False)
        e1.record()
        torch.cuda.synchronize()
        nms_ms = e0.elapsed_time(e1) / args.profile_passes
        plan_bytes = sum(a['bytes'] for a in conv)   # what the recorded (fused) launches need at minimum
        # ALGORITHMIC bytes: the graph's own layer list, every nn.Conv2d as one kernel (fp16 input + output + weights) -
        # the un-fused figure the graph's arithmetic intensity is quoted on - plus the decode's read + write of the head
        # tensor when the Detect kernels (conv + decode) are in the timed set.  Fusing layers lowers the time, not this.
        ref_bytes, ref_flops = reference_layer_work(model, x)
        conv_bytes = ref_bytes + (args.batch * DECODE_BYTES_PER_IMAGE_640 * (x.shape[2] * x.shape[3]) / (640.0 * 640.0) if det else 0.0)
        tflops = conv_flops / (conv_ms * 1e-3) / 1e12
        gbs = conv_bytes / (conv_ms * 1e-3) / 1e9
        # which roof binds the convolution work of this model: its FLOP/byte against the machine's
        hbm_bound = conv_bytes / (PEAK_HBM_GBS * 1e9) >= conv_flops / (PEAK_MFMA_TFLOPS * 1e12)
        # HBM bytes of the same launches from rocprofv3's counters (FETCH_SIZE x 2 + WRITE_SIZE, separate passes): they
        # cannot be collected inside this process, so the figure is the one tools/round_profile.sh measured for this
        # workload and committed, named with its source; null when no such file matches the workload
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, 'profiles', 'r3_hbm_traffic.json')
        if os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            if tj.get('model', 'yolov5s') == args.model and tj.get('batch', 32) == args.batch:
                traffic = tj.get('conv_total')
                traffic_source = 'profiles/r3_hbm_traffic.json: %s' % tj.get('source', 'tools/round_profile.sh, another process on another box')
        # B = 1 latency (detect.py's case): forward + NMS + host synchronisation per image, median of 30
        x1 = x[:1].contiguous()
        lat = []
        for i in range(0 if args.no_latency else 35):
            torch.cuda.synchronize()
            l0 = time.perf_counter()
            z1, _ = model(x1)
            hip_nms.nms_padded(z1, args.conf, args.iou, check_overflow=False)
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - l0) * 1e3)
        lat = sorted(lat[5:])
        latency_b1_ms = lat[len(lat) // 2] if lat else None
        top = sorted(conv, key=lambda a: -a['ms'])[:6]
        roofline = {'kernel': 'convolution kernels (stem2 / csp1 / conv_c64 / conv_c128 / conv_s2c64 / conv_p1x1 / conv_p1x1_cat / conv_ph / conv_igemm / conv_lw / conv_lin3 / conv_ws1x1 / conv3x3_ws / detect_p): all %d launches of one forward'
                              % len(conv),
                    'bound': 'hbm' if hbm_bound else 'mfma',
                    'achieved': round(gbs if hbm_bound else tflops, 2), 'peak': PEAK_HBM_GBS if hbm_bound else PEAK_MFMA_TFLOPS,
                    'unit': 'GB/s' if hbm_bound else 'TFLOP/s',
                    'frac': round(gbs / PEAK_HBM_GBS if hbm_bound else tflops / PEAK_MFMA_TFLOPS, 4), 'traffic': traffic,
                    'traffic_source': traffic_source,
                    'frac_plan_bytes': round(plan_bytes / (conv_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                    'bytes_per_forward': conv_bytes, 'bytes_basis': 'every Conv2d of the graph as one kernel (in + out + weights, fp16)' + (' + decode read/write of the head' if det else ''),
                    'plan_bytes_per_forward': plan_bytes, 'flops_per_forward': conv_flops, 'layer_list_flops': ref_flops, 'ms_per_forward': round(conv_ms, 4),
                    'mfma_tflops': round(tflops, 2), 'mfma_frac': round(tflops / PEAK_MFMA_TFLOPS, 4),
                    'per_op_event_scale': round(event_scale, 4),
                    'measured': 'per-launch event durations scaled to sum to the forward as the timed steps ran it' if pipe is None else
                                'per-launch event durations of ONE forward alone on the device (after the timed loop); in the timed loop %d steps are in flight and share the device, see in_flight' % args.in_flight}
        if pipe is not None:   # the same bytes over the whole step time of the timed loop: a lower bound of the kernels' rate there (the NMS kernels run inside it)
            roofline['in_flight'] = {'steps_in_flight': args.in_flight, 'ms_per_step': round(ms_step, 4),
                                     'achieved': round(conv_bytes / (ms_step * 1e-3) / 1e9, 2), 'unit': 'GB/s', 'frac': round(conv_bytes / (ms_step * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                                     'mfma_tflops': round(conv_flops / (ms_step * 1e-3) / 1e12, 2), 'mfma_frac': round(conv_flops / (ms_step * 1e-3) / 1e12 / PEAK_MFMA_TFLOPS, 4)}
        record = {
            'metric': 'images/sec at 640x640 fp16 (conv fwd+decode+NMS)', 'value': round(value, 1), 'unit': 'images/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_step, 4),
            'host_enqueue_ms_per_step': round(host_enqueue_ms, 4), 'forward_ms_in_loop': round(loop_forward_ms, 4),
            'latency_batch1_ms': None if latency_b1_ms is None else round(latency_b1_ms, 4), 'rank_ms_per_step': {'min': round(min(rank_ms), 4), 'max': round(max(rank_ms), 4)},
            'higher_is_better': True, 'scaling': 'strong' if args.global_batch else 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': '%s 640x640 %s batch %d per GPU, conf %g iou %g (test.py setting), random-init weights, '
                                   'BatchNorm-calibrated, head gain %g' % (args.model, 'fp16' if args.dtype == 'f16' else 'bf16', args.batch, args.conf, args.iou, args.head_gain),
                       'global_batch': global_batch, 'parallelism': 'batch shard x%d, detection all_gather (%s)' % (world, dist.get_backend() if world > 1 else 'none'),
                       'gather_ms': None if gather_ms is None else round(gather_ms, 4),
                       'tuning': 'kernel variants timed by this process' if world == 1 else 'kernel variants timed by rank 0, broadcast to the other ranks',
                       'cpus_per_rank': None if pinned is None else len(pinned),
                       'candidates_per_image': round(cand, 1), 'kept_per_image': round(kept, 1),
                       'steps_in_flight': args.in_flight if pipe is not None else 1,
                       'nms_stream': ('the step\'s own stream (%d steps in flight on %d streams, each with its own launch plan)' % (args.in_flight, args.in_flight)) if pipe is not None else 'same stream as the forward' if side is None else 'second stream: NMS of step i beside the forward of step i+1',
                       'outputs': 'fresh tensors every step' if post is None and pipe is None else 'two sets of output tensors reused in turn: no device allocation per step',
                       'forward_issue': 'one launch per kernel' if graphed is None else 'captured HIP graphs, one launch per forward'},
            'roofline': roofline,
            'stages': {'note': 'forward_ms and nms_ms are timed on their own after the timed loop (one stream, nothing overlapped); plan_ms / conv_ms / decode_ms are per-launch event durations scaled to sum to ' + ('forward_ms_in_loop' if pipe is None else 'forward_ms (forward_ms_in_loop is the duration of a forward that shares the device with the other steps in flight)') + '',
                       'forward_ms': round(forward_ms, 4), 'plan_ms': round(plan_ms, 4), 'conv_ms': round(conv_ms, 4), 'decode_ms': round(dec_ms, 4), 'decode_kernels': 'head conv + decode fused' if det else 'decode',
                       'nms_ms': round(nms_ms, 4),
                       'decode_hbm': {'bound': 'hbm', 'achieved': round(dec_bytes / (dec_ms * 1e-3) / 1e9, 1), 'peak': PEAK_HBM_GBS,
                                      'unit': 'GB/s', 'frac': round(dec_bytes / (dec_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)},
                       'top_convs': [{'label': a['label'], 'ms': round(a['ms'], 4),
                                      'tflops': round(a['flops'] / (a['ms'] * 1e-3) / 1e12, 1)} for a in top]},
        }
        if not args.no_cpu_baseline and world == 1:
            record['cpu_baseline'] = cpu_baseline(args.model, cpu_sd, strides, args.conf, args.iou)
        elif world == 1:
            record['cpu_baseline'] = None
        print(json.dumps(record))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
