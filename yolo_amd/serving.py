"""A serving / evaluation loop with several steps in flight.

The reference's loops (detect.py:63-73, test.py:164-182) run forward and NMS of one batch, then the next.  On this GPU
the small kernels of the 40x40 and 20x20 stages and the NMS leave compute units idle, and kernels of ANOTHER batch can
use them: Pipeline issues step k on stream k mod depth, each stream with a launch plan of its own (own activation
buffers, own outputs: engine.run_lane) and an NMS instance of its own (PaddedNms with a private workspace).  Every step
still runs the whole path on its batch; only the order in which the device executes the kernels of different steps is
free.  Measured on yolov5s at batch 32: 1.32 ms per step with one step in flight (NMS beside the next forward), 1.21
with two, 1.15 with three, 1.23 with four.

Nothing is allocated per step and the host never waits: a step returns its output tensors and an event; the tensors are
valid from that event on and are overwritten `depth * ring` steps later.
"""
import torch

from yolo_amd import _lib, engine
from yolo_amd.nms import MAX_DET, PaddedNms


class Step:
    """What Pipeline returns for one batch: det (B, max_det, 6) fp32, count (B,) int32, kept_index (B, max_det) int32,
    n_candidates (B,) int32, the raw outputs z and xs of the forward, the stream they were computed on and the event
    recorded behind them."""
    __slots__ = ('det', 'count', 'kept_index', 'n_candidates', 'z', 'xs', 'stream', 'done')

    def wait(self):
        """Make the current stream wait for this step's results (no host synchronisation)."""
        torch.cuda.current_stream(self.z.device).wait_event(self.done)
        return self

    def synchronize(self):
        self.done.synchronize()
        return self


class Pipeline:
    """pipeline = Pipeline(model, example_batch, depth=3, conf_thres=..., iou_thres=...); step = pipeline(batch).

    model: a yolo_amd Model the HIP engine accepts for `example_batch` (eval mode, half precision, on the GPU); every
    batch must have the example's shape, dtype and layout.  `batch` is read on the step's stream after the work already
    issued on the caller's current stream; the caller must leave it untouched until step.done (copy_input=True makes the
    pipeline copy it into a buffer of its own first, on the caller's stream: one more pass over the images, and the
    caller may reuse the batch tensor at once).
    nms=False leaves the NMS out (step.det etc. are None): forward only.
    Building a Pipeline switches the model's launch plans to ring outputs (engine.set_output_ring(model, ring)): model(x)
    called directly afterwards also returns tensors that are reused `ring` forwards later."""

    def __init__(self, model, example, depth=3, ring=2, conf_thres=0.1, iou_thres=0.6, classes=None, agnostic=False, merge=False,
                 max_det=MAX_DET, max_candidates=None, copy_input=False, nms=True):
        if depth < 1 or ring < 1:
            raise ValueError('depth and ring must be at least 1')
        self.model, self.depth, self.ring = model, int(depth), int(ring)
        self.device = example.device
        engine.set_output_ring(model, self.ring)
        with torch.cuda.device(self.device):
            out = engine.run_lane(model, example, 0)
            if out is None:
                raise _lib.Y5Error('Pipeline: this model / input does not run on the HIP engine')
            z = out[0]
            self.streams = [torch.cuda.Stream(self.device) for _ in range(self.depth)]
            self.posts = [PaddedNms(z.shape[0], z.shape[1], z.shape[2], z.dtype, self.device, conf_thres=conf_thres, iou_thres=iou_thres,
                                    classes=classes, agnostic=agnostic, max_det=max_det, max_candidates=max_candidates, ring=self.ring,
                                    merge=merge, private_workspace=self.depth > 1) for _ in range(self.depth)] if nms else None
            for lane in range(1, self.depth):   # builds the other lanes' plans now (kernel choices are replayed, not timed again)
                engine.run_lane(model, example, lane)
            self.inputs = [torch.empty_like(example) for _ in range(self.depth)] if copy_input else None
            self.events = [[torch.cuda.Event() for _ in range(self.ring)] for _ in range(self.depth)]
            self.steps = [[Step() for _ in range(self.ring)] for _ in range(self.depth)]
            torch.cuda.synchronize(self.device)
        self.shape, self.dtype = tuple(example.shape), example.dtype
        self.issued = 0
        self.last = [None] * self.depth   # the event behind each lane's latest step

    def __call__(self, x, timing=None):
        """Issue one step for batch x; returns its Step at once.  timing: a pair of timing events, recorded on the step's
        stream right before and right after the forward (benchmarks)."""
        if tuple(x.shape) != self.shape or x.dtype is not self.dtype or x.device != self.device:
            raise ValueError('Pipeline was built for a %s %s batch on %s' % (self.shape, self.dtype, self.device))
        k = self.issued
        lane, slot = k % self.depth, (k // self.depth) % self.ring
        self.issued = k + 1
        s = self.streams[lane]
        cur = torch.cuda.current_stream(self.device)
        if self.inputs is not None:
            # the copy runs on the caller's stream (so the caller may reuse x as soon as this returns), after the lane's
            # previous step, which reads the same buffer
            if k >= self.depth:
                cur.wait_event(self.last[lane])
            self.inputs[lane].copy_(x)
            x = self.inputs[lane]
        s.wait_stream(cur)
        if timing is not None:
            timing[0].record(s)
        out = engine.run_lane(self.model, x, lane, s)
        if out is None:
            raise _lib.Y5Error('Pipeline: the model no longer runs on the HIP engine (training mode, hooks, another dtype?)')
        step = self.steps[lane][slot]
        step.z, step.xs = out
        if timing is not None:
            timing[1].record(s)
        if self.posts is not None:
            step.det, step.count, step.kept_index, step.n_candidates = self.posts[lane](step.z, stream=s, after_current=False)
        else:
            step.det = step.count = step.kept_index = step.n_candidates = None
        step.stream, step.done = s, self.events[lane][slot]
        step.done.record(s)
        self.last[lane] = step.done
        return step

    def synchronize(self):
        for s in self.streams:
            s.synchronize()
