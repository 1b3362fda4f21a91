"""BatchNorm statistics recalibration ("adaptive BN") of a pruned candidate on the HIP kernels.

What replaces what (reference call sites):
  * Training.adaptive_BN (filter_pruning/fp_train.py:207-261): `model.train()`, then under torch.no_grad() one forward
    per loader batch of `inputs.float() / 255.0`.  Nothing is returned by those forwards; their only effect is that every
    nn.BatchNorm2d (Conv.bn models/common.py:27, BottleneckCSP.bn models/common.py:62) normalises with the batch's own
    statistics and moves running_mean / running_var by its momentum (0.03, utils/torch_utils.py:69).
  * the search loop that calls it for every candidate (filter_pruning/02-search_filter_pruning_.py:93-143).

Here each batch runs as: convolution without post-ops that also accumulates the per-channel sums of its output
(y5_conv_stats_f16; or y5_conv_f16 + a pass of y5_bn_stats_f16 where that measures faster) -> normalisation coefficients + running-statistics update (y5_bn_finalize_f32) -> normalise + LeakyReLU (+ Bottleneck
shortcut) in place (y5_affine_act_f16).  Activations are NHWC fp16 (weights are rounded to fp16, sums and statistics
are fp32 / fp64), so the statistics agree with the reference's fp32 forward to fp16 accuracy, not bit for bit; the
tests state the tolerance.  The Detect layer has no BatchNorm and its train-mode output is discarded by adaptive_BN,
so its convolutions are not run.

The running statistics live in padded device vectors while batches are processed and are written back to the
modules (with num_batches_tracked) by Recalibrator.finish().
"""
import ctypes as C
from copy import deepcopy

import torch
import torch.nn as nn

from yolo_amd import _lib
from yolo_amd.engine import Act, Unsupported, _conv_geometry, _TUNED, pack_conv_weight, pad8


class _BN:
    """One nn.BatchNorm2d in the padded channel layout `segs` of the tensor it normalises."""

    def __init__(self, bn, segs, device):
        if not isinstance(bn, nn.BatchNorm2d):
            raise Unsupported('normalisation %s' % type(bn).__name__)
        if sum(c for c, _ in segs) != bn.num_features:
            raise Unsupported('BatchNorm width %d on a tensor of %d channels' % (bn.num_features, sum(c for c, _ in segs)))
        self.bn = bn
        self.c_pad = sum(p for _, p in segs)
        idx, off = [], 0
        for c, p in segs:
            idx.extend(range(off, off + c))
            off += p
        self.idx = torch.tensor(idx, dtype=torch.long, device=device)
        z = lambda fill=0.0: torch.full((self.c_pad,), fill, dtype=torch.float32, device=device)
        self.gamma, self.beta = z(), z()   # zero in the padding channels: their outputs stay zero
        self.gamma[self.idx] = bn.weight.detach().float() if bn.affine else 1.0
        if bn.affine:
            self.beta[self.idx] = bn.bias.detach().float()
        self.track = bn.track_running_stats and bn.running_mean is not None
        self.rm, self.rv = z(), z(1.0)
        if self.track:
            self.rm[self.idx] = bn.running_mean.detach().float()
            self.rv[self.idx] = bn.running_var.detach().float()
        self.scale, self.shift = z(), z()
        self.partials = {}   # pixels -> fp32 workspace of the per-workgroup partial sums
        self.tracked0 = int(bn.num_batches_tracked) if self.track and bn.num_batches_tracked is not None else 0
        self.batches = 0

    def momentum(self):
        """nn.BatchNorm2d's exponential factor for the batch about to be seen (cumulative average when momentum is None)."""
        if self.bn.momentum is None:
            return 1.0 / float(self.tracked0 + self.batches + 1)
        return float(self.bn.momentum)

    def write_back(self):
        if self.track and self.batches:
            self.bn.running_mean.copy_(self.rm[self.idx].to(self.bn.running_mean.dtype))
            self.bn.running_var.copy_(self.rv[self.idx].to(self.bn.running_var.dtype))
            if self.bn.num_batches_tracked is not None:
                self.bn.num_batches_tracked += self.batches
        self.tracked0 += self.batches
        self.batches = 0


class _Program:
    """The kernel calls of one train-mode forward for a fixed input shape, with the buffers they use."""

    def __init__(self, owner, shape, u8):
        self.owner, self.lib, self.device = owner, owner.lib, owner.device
        self.batch, self.cin, self.h, self.w = shape
        self.u8 = u8
        self.keep, self.ops, self.labels = [], [], []
        self.image = torch.zeros(shape, dtype=torch.float16, device=self.device)   # the batch as fp16 NCHW
        # partial sums that convolutions accumulate for the BatchNorms behind them: one arena, zeroed once per batch
        total = sum(m.num_features + 16 for m in owner.model.modules() if isinstance(m, nn.BatchNorm2d))
        self.arena = torch.zeros(self.STAT_ROWS * 2 * total, dtype=torch.float32, device=self.device)
        self.arena_used = 0
        self._build(owner.model)

    # ------------------------------------------------------------------ helpers
    def _new(self, h, w, segs):
        c = sum(p for _, p in segs)
        buf = torch.zeros(self.batch * h * w * c, dtype=torch.float16, device=self.device)
        self.keep.append(buf)
        return Act(buf, 0, c, self.batch, h, w, segs)

    def _call(self, label, fn, *args):
        self.ops.append((fn, args))
        self.labels.append(label)

    STAT_ROWS = 128   # rows of partial sums a convolution's epilogue spreads its per-tile sums over

    def _conv_desc(self, x, conv, out):
        """Descriptor of conv2d without bias, normalisation or activation: out = fp16(conv(x))."""
        k, s = _conv_geometry(conv)
        if conv.bias is not None:
            raise Unsupported('convolution with bias in front of a BatchNorm')
        ho, wo = (x.h + 2 * (k // 2) - k) // s + 1, (x.w + 2 * (k // 2) - k) // s + 1
        if (ho, wo) != (out.h, out.w) or len(out.segs) != 1 or out.segs[0][0] != conv.out_channels:
            raise Unsupported('conv output slice mismatch')
        packed, kpad, rows = pack_conv_weight([(conv.weight, out.segs[0][1])], x.segs, self.device)
        self.keep.append(packed)
        d = _lib.ConvDesc()
        d.x, d.w, d.y = x.ptr, packed.data_ptr(), out.ptr
        d.batch, d.h, d.w_in, d.cin, d.x_ld = x.batch, x.h, x.w, x.c_pad, x.ld
        d.ho, d.wo, d.cout, d.y_ld = ho, wo, out.c_pad, out.ld
        d.ksize, d.stride, d.pad = k, s, k // 2
        d.kpad, d.w_rows = kpad, rows
        d.act, d.slope = _lib.ACT_NONE, 0.0
        self.keep.append(d)
        return d

    def _time(self, call):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        call()
        e0.record()
        for _ in range(3):
            call()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / 3

    def _tune(self, d, x):
        """(variant, ms) of the fastest tile configuration for the plain convolution and of the fastest one that also
        accumulates the output's per-channel sums, timed on the real shape (cached per process)."""
        if not self.owner.autotune:
            return (0, 0.0), (0, 0.0)
        key = ('abn', d.batch, d.h, d.w_in, d.cin, d.x_ld, d.cout, d.y_ld, d.ksize, d.stride)
        if key not in _TUNED:
            x.buf.uniform_(-1.0, 1.0)
            stream = _lib.current_stream()
            scratch = torch.zeros(self.STAT_ROWS * 2 * d.cout, dtype=torch.float32, device=self.device)
            plain, fused = (0, float('inf')), (0, float('inf'))
            for v in range(self.lib.y5_conv_num_variants() + 1):
                d.variant = v
                if self.lib.y5_conv_f16(C.byref(d), stream) != 0:
                    continue  # configuration does not apply to this shape
                t = self._time(lambda: self.lib.y5_conv_f16(C.byref(d), stream))
                if t < plain[1]:
                    plain = (v, t)
                if v and self.lib.y5_conv_stats_f16(C.byref(d), scratch.data_ptr(), self.STAT_ROWS, d.cout, 0, stream) == 0:
                    t = self._time(lambda: self.lib.y5_conv_stats_f16(C.byref(d), scratch.data_ptr(), self.STAT_ROWS, d.cout, 0, stream))
                    if t < fused[1]:
                        fused = (v, t)
            _TUNED[key] = (plain, fused)
        return _TUNED[key]

    def _fuse_stats(self, d, x):
        """Whether this convolution should accumulate the statistics itself: when its best accumulating configuration costs
        less than the best plain one plus a separate pass over the output (>= 6 us, ~3 TB/s)."""
        if not self.owner.fuse_stats:
            return False
        plain, fused = self._tune(d, x)
        if not self.owner.autotune:
            return True
        stats_ms = max(6e-3, d.batch * d.ho * d.wo * d.cout * 2 / 3.0e9)
        return fused[1] <= plain[1] + stats_ms

    def _emit_raw(self, label, d, x, stat=None):
        """Record the convolution; stat = (partials pointer, rows, total channels, channel offset) makes it accumulate."""
        plain, fused = self._tune(d, x)
        d.variant = fused[0] if stat else plain[0]
        text = '%s conv%dx%d s%d %d->%d @%dx%d v%d' % (label, d.ksize, d.ksize, d.stride, x.c_log, d.cout, d.ho, d.wo, d.variant)
        if stat:
            self._call(text + ' +sums', self.lib.y5_conv_stats_f16, C.byref(d), *stat)
        else:
            self._call(text, self.lib.y5_conv_f16, C.byref(d))

    def _stat_slice(self, c_pad):
        n = self.STAT_ROWS * 2 * c_pad
        if self.arena_used + n > self.arena.numel():
            raise RuntimeError('adaptive_bn: statistics arena too small')
        part = self.arena[self.arena_used:self.arena_used + n]
        self.arena_used += n
        return part

    def _batchnorm(self, label, bn, t, slope, res=None, part=None):
        """Train-mode BatchNorm2d + optional LeakyReLU (+ residual) over tensor t, in place.  part: the partial sums the
        producing convolutions accumulated (STAT_ROWS rows); None = a pass over t computes them."""
        st = self.owner.bn_state(bn, t.segs)
        pixels = t.batch * t.h * t.w
        if part is None:
            rows = self.lib.y5_bn_stats_rows(pixels, t.c_pad)
            if rows <= 0:
                raise Unsupported('BatchNorm over %d channels' % t.c_pad)
            part = st.partials.get(pixels)
            if part is None:
                part = st.partials[pixels] = torch.zeros(rows * 2 * t.c_pad, dtype=torch.float32, device=self.device)
            self._call(label + ' bn stats', self.lib.y5_bn_stats_f16, t.ptr, t.ld, pixels, t.c_pad, part.data_ptr())
        else:
            rows = self.STAT_ROWS
        self.ops.append((self._finalize, (st, part, rows, pixels)))
        self.labels.append(label + ' bn finalize')
        act = _lib.ACT_NONE if slope is None else _lib.ACT_LEAKY
        self._call(label + ' bn apply', self.lib.y5_affine_act_f16, t.ptr, t.ld, t.ptr, t.ld, None if res is None else res.ptr,
                   0 if res is None else res.ld, pixels, t.c_pad, st.scale.data_ptr(), st.shift.data_ptr(), act,
                   0.0 if slope is None else slope)

    def _finalize(self, st, part, rows, pixels, stream):
        rc = self.lib.y5_bn_finalize_f32(part.data_ptr(), rows, pixels, st.c_pad, st.c_pad, st.gamma.data_ptr(), st.beta.data_ptr(),
                                         float(st.bn.eps), st.momentum(), st.rm.data_ptr() if st.track else None,
                                         st.rv.data_ptr() if st.track else None, st.scale.data_ptr(), st.shift.data_ptr(), stream)
        st.batches += 1
        return rc

    @staticmethod
    def _slope(act):
        if act is None or isinstance(act, nn.Identity):
            return None
        if isinstance(act, nn.LeakyReLU):
            return float(act.negative_slope)
        raise Unsupported('activation %s' % type(act).__name__)

    def _conv_block(self, label, m, x, out, res=None):
        """models.common.Conv in train mode: act(bn(conv(x))) (models/common.py:26-27)."""
        if m.bn is None:
            raise Unsupported('Conv with its BatchNorm folded away (Model.fuse) has no statistics to recalibrate')
        if res is not None and (res.c_pad, res.h, res.w) != (out.c_pad, out.h, out.w):
            raise Unsupported('residual shape mismatch')
        d = self._conv_desc(x, m.conv, out)
        part = self._stat_slice(out.c_pad) if self._fuse_stats(d, x) else None
        self._emit_raw(label, d, x, None if part is None else (part.data_ptr(), self.STAT_ROWS, out.c_pad, 0))
        self._batchnorm(label, m.bn, out, self._slope(m.act), res, part)

    def _block_out(self, x, conv, s=1):
        c = conv.out_channels
        return self._new((x.h - 1) // s + 1, (x.w - 1) // s + 1, [(c, pad8(c))])

    def _bottleneck(self, label, m, x):
        hidden = self._block_out(x, m.cv1.conv)
        self._conv_block(label + '.cv1', m.cv1, x, hidden)
        out = self._block_out(hidden, m.cv2.conv)
        self._conv_block(label + '.cv2', m.cv2, hidden, out, res=x if m.add else None)
        return out

    def _csp(self, label, m, x):
        """BottleneckCSP.forward (models/common.py:59-62): cv3 and cv2 write the two halves of one buffer, whose
        BatchNorm + LeakyReLU then run in place."""
        t = self._block_out(x, m.cv1.conv)
        self._conv_block(label + '.cv1', m.cv1, x, t)
        for bi, b in enumerate(m.m):
            t = self._bottleneck('%s.m%d' % (label, bi), b, t)
        c3, c2 = m.cv3.out_channels, m.cv2.out_channels
        cat = self._new(x.h, x.w, [(c3, pad8(c3)), (c2, pad8(c2))])
        d3 = self._conv_desc(t, m.cv3, cat.slice(0, [(c3, pad8(c3))]))
        d2 = self._conv_desc(x, m.cv2, cat.slice(pad8(c3), [(c2, pad8(c2))]))
        part = self._stat_slice(cat.c_pad) if self._fuse_stats(d3, t) and self._fuse_stats(d2, x) else None
        stat = (lambda off: None) if part is None else (lambda off: (part.data_ptr(), self.STAT_ROWS, cat.c_pad, off))
        self._emit_raw(label + '.cv3', d3, t, stat(0))
        self._emit_raw(label + '.cv2', d2, x, stat(pad8(c3)))
        self._batchnorm(label, m.bn, cat, self._slope(m.act), part=part)
        out = self._block_out(cat, m.cv4.conv)
        self._conv_block(label + '.cv4', m.cv4, cat, out)
        return out

    def _spp(self, label, m, x):
        """SPP.forward (models/common.py:74-76)."""
        c = m.cv1.conv.out_channels
        seg = [(c, pad8(c))]
        cat = self._new(x.h, x.w, seg * (len(m.m) + 1))
        first = cat.slice(0, seg)
        self._conv_block(label + '.cv1', m.cv1, x, first)
        for n, p in enumerate(m.m):
            k = p.kernel_size if isinstance(p.kernel_size, int) else p.kernel_size[0]
            st = p.stride if isinstance(p.stride, int) else p.stride[0]
            pd = p.padding if isinstance(p.padding, int) else p.padding[0]
            if st != 1 or pd != k // 2 or k % 2 != 1 or k > 13 or p.dilation != 1 or p.ceil_mode:
                raise Unsupported('SPP pool k%d s%d p%d' % (k, st, pd))
            dst = cat.slice((n + 1) * pad8(c), seg)
            self._call('%s maxpool%d' % (label, k), self.lib.y5_maxpool_f16, first.ptr, first.ld, dst.ptr, dst.ld, x.batch, x.h, x.w,
                       pad8(c), k)
        out = self._block_out(cat, m.cv2.conv)
        self._conv_block(label + '.cv2', m.cv2, cat, out)
        return out

    def _build(self, model):
        from yolo_amd.models import common, yolo
        B, H, W = self.batch, self.h, self.w
        outs = []
        for i, m in enumerate(model.model):
            label = '%d.%s' % (i, type(m).__name__)
            f = m.f

            def get(j):
                return outs[i + j if j < 0 else j]
            if isinstance(m, yolo.Detect):
                outs.append(None)   # no BatchNorm inside, and adaptive_BN drops the model's output
                continue
            if i == 0:
                if f != -1 or not isinstance(m, (common.Focus, common.Conv)):
                    raise Unsupported('first layer %s' % type(m).__name__)
                focus = isinstance(m, common.Focus)
                if focus and (H % 2 or W % 2):
                    raise Unsupported('Focus on an odd image size')
                c = 4 * self.cin if focus else self.cin
                x = self._new(H // 2 if focus else H, W // 2 if focus else W, [(c, pad8(c))])
                self._call(label + ' pack', self.lib.y5_pack_image_f16, self.image.data_ptr(), x.ptr, B, self.cin, H, W,
                           1 if focus else 0, x.ld)
                block = m.conv if focus else m
                s = _conv_geometry(block.conv)[1]
                out = self._block_out(x, block.conv, s)
                self._conv_block(label, block, x, out)
            elif isinstance(m, common.Conv):
                x = get(f)
                out = self._block_out(x, m.conv, _conv_geometry(m.conv)[1])
                self._conv_block(label, m, x, out)
            elif isinstance(m, common.Bottleneck):
                out = self._bottleneck(label, m, get(f))
            elif isinstance(m, common.BottleneckCSP):
                out = self._csp(label, m, get(f))
            elif isinstance(m, common.SPP):
                out = self._spp(label, m, get(f))
            elif isinstance(m, nn.Upsample):
                x = get(f)
                sf = m.scale_factor if isinstance(m.scale_factor, (tuple, list)) else (m.scale_factor, m.scale_factor)
                if m.size is not None or tuple(float(v) for v in sf) != (2.0, 2.0) or m.mode not in ('nearest', 'bilinear') or \
                        (m.mode == 'bilinear' and m.align_corners):
                    raise Unsupported('upsample %s x%s' % (m.mode, m.scale_factor))
                out = self._new(2 * x.h, 2 * x.w, x.segs)
                self._call(label + ' upsample2x', self.lib.y5_upsample2x_f16, x.ptr, x.ld, out.ptr, out.ld, B, x.h, x.w, x.c_pad,
                           0 if m.mode == 'nearest' else 1)
            elif isinstance(m, common.Concat):
                parts = [get(j) for j in f]
                if m.d != 1 or len({(p.h, p.w) for p in parts}) != 1:
                    raise Unsupported('concat')
                out = self._new(parts[0].h, parts[0].w, [sg for p in parts for sg in p.segs])
                off = 0
                for p in parts:
                    self._call(label + ' copy', self.lib.y5_copy_channels_f16, p.ptr, p.ld, out.ptr + 2 * off, out.ld,
                               B * p.h * p.w, p.c_pad)
                    off += p.c_pad
            else:
                raise Unsupported('module %s' % type(m).__name__)
            outs.append(out)
        self.outs = outs

    def run(self, inputs):
        stream = _lib.current_stream()
        if self.u8:
            _lib.check(self.lib.y5_normalize_u8_f16(inputs.data_ptr(), self.image.data_ptr(), inputs.numel(), stream))
        else:
            self.image.copy_(inputs)
        if self.arena_used:
            self.arena[:self.arena_used].zero_()
        for (fn, args), label in zip(self.ops, self.labels):
            rc = fn(*args, stream)
            if rc != 0:
                _lib.check(rc)


class Recalibrator:
    """Runs train-mode forwards of `model` (a models.yolo.Model on a GPU) on the HIP kernels and keeps the moved BatchNorm
    statistics; finish() writes them into the modules."""

    def __init__(self, model, autotune=True, fuse_stats=True):
        """autotune: time the compiled tile configurations per layer; fuse_stats: let convolutions accumulate the sums of
        their output where that is faster than a separate pass."""
        self.model = model
        self.fuse_stats = fuse_stats
        self.lib = _lib.lib()
        self.device = next(model.parameters()).device
        if self.device.type != 'cuda':
            raise RuntimeError('adaptive_bn: the model must be on a GPU (the HIP kernels have no CPU path)')
        self.autotune = autotune
        self._bn = {}
        self._programs = {}

    def bn_state(self, bn, segs):
        st = self._bn.get(id(bn))
        if st is None:
            st = self._bn[id(bn)] = _BN(bn, segs, self.device)
        elif st.c_pad != sum(p for _, p in segs):
            raise Unsupported('one BatchNorm on tensors of different layouts')
        return st

    def step(self, inputs):
        """One batch: uint8 (divided by 255 like `inputs.float() / 255.0`) or float, NCHW, on the model's device."""
        if inputs.dim() != 4 or inputs.device != self.device:
            raise ValueError('adaptive_bn: expected an NCHW batch on %s' % self.device)
        inputs = inputs.contiguous()
        key = (tuple(inputs.shape), inputs.dtype is torch.uint8)
        prog = self._programs.get(key)
        if prog is None:
            prog = self._programs[key] = _Program(self, key[0], key[1])
        prog.run(inputs)

    def finish(self):
        torch.cuda.current_stream(self.device).synchronize()
        for st in self._bn.values():
            st.write_back()
        return self.model


def adaptive_bn(in_model, dataloader, deepcp=True, device=None):
    """Training.adaptive_BN (filter_pruning/fp_train.py:207-261) on the HIP kernels: returns the model (a deep copy unless
    deepcp is False), left in train() mode like the reference leaves it, with recalibrated BatchNorm statistics.
    dataloader yields (inputs, targets, paths, shapes) tuples like the reference's loader, or bare image batches."""
    model = deepcopy(in_model) if deepcp else in_model
    model.train()
    rec = Recalibrator(model)
    device = device or rec.device
    done = 0
    with torch.no_grad():
        it = iter(dataloader)
        for batch in it:
            inputs = batch[0] if isinstance(batch, (tuple, list)) else batch
            inputs = inputs.to(device, non_blocking=True)
            if inputs.dtype is not torch.uint8:
                # the reference divides every batch by 255 whatever its dtype (fp_train.py:239); Recalibrator.step only
                # does so for uint8 and takes a float batch as already normalised
                inputs = inputs.float() / 255.0
            try:
                rec.step(inputs)
            except Unsupported:
                if done:
                    raise  # statistics already moved by the kernels: cannot hand the same batches to torch again
                # a model the kernels do not take (grouped or biased convolutions, fused Conv modules ...): the reference's
                # own loop on the torch modules, this batch included
                import itertools
                for b in itertools.chain([batch], it):
                    x = b[0] if isinstance(b, (tuple, list)) else b
                    model(x.to(device, non_blocking=True).float() / 255.0)
                return model
            done += 1
    return rec.finish()
