"""Lowers a live models.yolo.Model into a launch plan of HIP kernels and runs it.

What replaces what (reference call sites):
  * Model.forward_once's module walk (models/yolo.py:107-130)      -> one y5_plan_run call
  * Conv / fused Conv, Bottleneck add, BottleneckCSP incl. its concat BatchNorm,
    SPP, Focus, nn.Upsample + Concat (models/common.py)            -> conv / pack / maxpool / upsample ops
  * Detect head conv + decode (models/yolo.py:24-46)                -> conv + decode ops

Data layout: every activation is NHWC fp16 in a private buffer; channel counts are padded to
multiples of 8 with zero channels, so filter-pruned models with odd widths keep 16-byte vectors.
A concatenation is a wider buffer whose producers write their channel slice directly; a tensor's
`segs` list records (logical, padded) channel runs so that consumer weights can be packed to match.
Shapes and parameters are read from the live modules (weight.shape, not the yaml), which is what
makes pruned models work, and the plan is rebuilt whenever parameters change identity or version.
"""
import collections
import ctypes as C
import json
import math
import operator
import os
import weakref

import torch
import torch.nn as nn
import torch.nn.modules.module as _tm

from yolo_amd import _lib
from yolo_amd.utils.torch_utils import bn_affine

_DISABLED = [0]
_VERSION_OF = operator.attrgetter('_version')  # the two per-tensor reads of Engine.state(), as C callables for map()
_ADDRESS_OF = torch.Tensor.data_ptr
def _fastest(candidates, launch):
    """The fastest of `candidates` (variant ids) and its milliseconds per launch.  launch(v) issues one launch of variant v
    on the current stream and returns 0, or non-zero when the variant does not apply to the shape.  Every applicable
    variant is timed over four launches; the three fastest are within timing noise of each other after that, so they are
    timed again over twelve launches in two interleaved rounds (a clock or cache transient then hits all of them alike)
    and each keeps its better round.  Returns (None, inf) when no candidate applies."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    timed = []
    for v in candidates:
        if launch(v) != 0:
            continue
        e0.record()
        for _ in range(4):
            launch(v)
        e1.record()
        e1.synchronize()
        timed.append((e0.elapsed_time(e1) / 4, v))
    if not timed:
        return None, float('inf')
    again = {v: float('inf') for _, v in sorted(timed)[:3]}
    for _ in range(2):
        for v in again:
            e0.record()
            for _ in range(12):
                launch(v)
            e1.record()
            e1.synchronize()
            again[v] = min(again[v], e0.elapsed_time(e1) / 12)
    best = min(again, key=again.get)
    return best, again[best]


_TUNED = {}   # conv shape key -> fastest variant id, shared by all plans of the process
_TUNED_MS = {}  # key -> milliseconds per launch of the chosen variant
_ENGINES = weakref.WeakKeyDictionary()  # model -> Engine; kept out of the module so that deepcopy / pickle / torch.save of
#                                         a model that has run on the engine see no ctypes handles
MAX_PLANS = int(os.environ.get('Y5_MAX_PLANS', '4'))  # launch plans (one per input shape) an Engine keeps; least recently used goes first


def _cache_key(key):
    return json.dumps(key)


_REPLAYED = set()  # keys whose choice came from a file and has not been launched by this process yet


def _fingerprint():
    """What a variant id means depends on the library build (ids are positions in its kernel tables) and on the device:
    a cache written under another build or architecture is ignored."""
    import hashlib
    lib = _lib.lib()
    h = hashlib.sha1()
    with open(_lib.LIB_PATH, 'rb') as f:
        for block in iter(lambda: f.read(1 << 20), b''):
            h.update(block)
    arch = C.create_string_buffer(64)
    if lib.y5_device_arch(arch, 64) != 0:
        arch.value = b'unknown'
    return {'library': h.hexdigest(), 'arch': arch.value.decode(), 'version': lib.y5_version(),
            'variants': [lib.y5_conv_num_variants(), lib.y5_conv_num_variants_bf16(), lib.y5_conv_chain_num_variants(),
                         lib.y5_conv_cat2_num_variants(), lib.y5_detect_num_variants()]}


def export_tuning():
    """Every variant choice of this process as plain data: {json(shape key): [variant, ms]} plus the library fingerprint
    under '__fingerprint__' (what save_tune_cache writes, what ranks send each other)."""
    data = {_cache_key(list(k)): [v, _TUNED_MS.get(k, 0.0)] for k, v in _TUNED.items()}
    data['__fingerprint__'] = _fingerprint()
    return data


def import_tuning(data):
    """Take the choices of export_tuning(): later plans of those shapes use them without timing anything (each is
    launched once before it is recorded; one the library rejects is timed afresh).  Returns the number of choices
    taken: 0 when they were made under another library build or device architecture."""
    data = dict(data)
    if data.pop('__fingerprint__', None) != _fingerprint():
        return 0
    for k, (variant, ms) in data.items():
        key = tuple(json.loads(k))
        _TUNED[key], _TUNED_MS[key] = int(variant), float(ms)
        _REPLAYED.add(key)
    return len(data)


def load_tune_cache(path):
    """import_tuning() of a file written by save_tune_cache."""
    with open(path) as f:
        return import_tuning(json.load(f))


def save_tune_cache(path):
    """Write export_tuning() as JSON."""
    data = export_tuning()
    tmp = '%s.%d.tmp' % (path, os.getpid())
    with open(tmp, 'w') as f:
        json.dump(data, f, indent=0, sort_keys=True)
    os.replace(tmp, path)
    return len(data) - 1


def _replay_ok(key, launch):
    """True when `key` has a usable choice.  A choice read from a file is launched once first (launch(variant) -> status);
    if the library rejects it, it is dropped so that the caller times the shape afresh."""
    if key not in _TUNED:
        return False
    if key in _REPLAYED:
        _REPLAYED.discard(key)
        if launch(_TUNED[key]) != 0:
            del _TUNED[key]
            _TUNED_MS.pop(key, None)
            return False
    return True


def _tune_cache_init():
    """Y5_TUNE_CACHE=<file>: choices are read from it at import (when it exists) and written back at exit, so that a
    second process - a profiler run, a test that wants the same kernels twice - replays the first one's plan."""
    path = os.environ.get('Y5_TUNE_CACHE')
    if not path:
        return
    if os.path.exists(path):
        try:
            load_tune_cache(path)
        except (ValueError, OSError, _lib.Y5Error) as e:
            import warnings
            warnings.warn('Y5_TUNE_CACHE %s ignored: %s' % (path, e))
    import atexit
    atexit.register(lambda: _TUNED and save_tune_cache(path))


_tune_cache_init()


class disabled:
    """Context manager that forces Model.forward onto the plain torch module walk (used by parity tests)."""

    def __enter__(self):
        _DISABLED[0] += 1

    def __exit__(self, *exc):
        _DISABLED[0] -= 1


_PAD_REGULAR = [False]  # while a Plan is being built: pad channel counts to regular widths instead of multiples of 8


def pad8(c):
    """Channels a tensor of c logical channels occupies in memory: the next multiple of 8 (16-byte vectors), or - for the
    plans of a model with irregular widths that measured faster that way - the next regular width: 32, 64, 128, ... when
    that is within a third of c, else the next multiple of 32.  The extra channels hold zeros (zero weight rows, zero bias);
    the kernels written for regular widths (persistent 1x1, weights-in-registers chains, whole blocks) then apply to the
    channel plans filter pruning leaves behind (filter_pruning/torch_pruning/prune/structured.py:48-60: 13, 25, 55, 119 ...)."""
    if _PAD_REGULAR[0] and c > 16:
        for w in (32, 64, 128, 256, 512, 1024):
            if c <= w <= c * 1.34:
                return w
        return (c + 31) // 32 * 32
    return (c + 7) // 8 * 8


def _irregular_widths(model):
    """True when some convolution of the model has a channel count that is not a multiple of 16 (the Detect convolutions
    and the image side of the first layer aside): a pruned channel plan.  The shipped graphs, yolov5m's 48 and yolov5x's 80
    channels included, are regular in this sense and are planned once."""
    from yolo_amd.models.yolo import Detect
    head = model.model[-1] if isinstance(model.model[-1], Detect) else None
    skip = set(id(m) for m in head.m) if head is not None else set()
    for m in model.modules():
        if isinstance(m, nn.Conv2d) and id(m) not in skip and (m.out_channels % 16 or (m.in_channels % 16 and m.in_channels > 16)):
            return True
    return False


def _supported_types():
    from yolo_amd.models import common, yolo
    return (common.Conv, common.Bottleneck, common.BottleneckCSP, common.SPP, common.Focus, common.Concat,
            yolo.Detect, nn.Upsample, nn.Sequential)


_REG_EPOCH = [0]  # bumped whenever a parameter, buffer or sub-module is assigned to ANY module of the process


def _registered(*_):
    _REG_EPOCH[0] += 1


# nn.Module.__setattr__ / register_* run these for every assignment of a Parameter, a buffer or a sub-module: pruning
# (filter_pruning/torch_pruning/prune/structured.py:50-60 assigns new Parameters), Model.fuse (models/yolo.py:155-163
# replaces m.conv and clears m.bn) and model surgery in general are seen without walking the model on every forward.
_tm.register_module_parameter_registration_hook(_registered)
_tm.register_module_buffer_registration_hook(_registered)
_tm.register_module_module_registration_hook(_registered)


def _has_hooks(model):
    if _tm._global_forward_hooks or _tm._global_forward_pre_hooks:
        return True
    for m in model.modules():
        if m._forward_hooks or m._forward_pre_hooks:
            return True
    return False


def input_layout(x):
    """How the image bytes of a 4-d (batch, channel, height, width) input lie in memory: _lib.IN_U8_HWC(_BGR) for a uint8
    tensor that is a permuted view of a (batch, height, width, 3) array - torch's channels_last - as
    yolo_amd.preprocess.from_hwc returns it; 0 for everything else (read as, or copied to, contiguous NCHW).
    An interleaved uint8 view WITHOUT from_hwc's marking is rejected: slicing, .to(device), pin_memory or a collate
    function return plain tensors, and guessing the channel order of those would silently swap red and blue."""
    if (x.dtype is torch.uint8 and x.dim() == 4 and x.shape[1] == 3 and not x.is_contiguous()
            and x.is_contiguous(memory_format=torch.channels_last)):
        bgr = getattr(x, 'y5_bgr', None)
        if bgr is None:
            raise TypeError('an interleaved (height, width, 3) uint8 batch must be passed through yolo_amd.preprocess.from_hwc '
                            'right before the model call: this view does not say whether its channels are BGR or RGB '
                            '(tensor operations such as slicing or .to(device) drop that marking)')
        return _lib.IN_U8_HWC_BGR if bgr else _lib.IN_U8_HWC
    return 0


def _plan_if_eligible(model, x):
    """The launch plan that serves this forward, or None when it has to run on the torch modules: inference mode on a
    GPU in fp16 or bf16, no autograd, no module hooks (the pruning / FLOP-counting tools attach hooks and need the torch
    modules to run), Detect.export off, 4-d input whose height and width are multiples of the largest stride, every
    top-level module lowered.  In the steady state (same model, same input shape) this costs one pass over the cached
    tensor list (version counters and addresses) and one over the cached hook tables: no module walk."""
    if _DISABLED[0] or model.training or torch.is_grad_enabled():
        return None
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 4):
        return None
    eng = _engine_of(model)
    st = eng.state()
    if x.dtype is torch.uint8:
        # the data loader's tensor (utils/datasets.py:140-141): the engine applies img.half() / 255.0 (test.py:152-154)
        # itself, inside the first kernel where it can; only meaningful for a half-precision model
        if st.dtype is not torch.float16:
            return None
    elif x.dtype not in (torch.float16, torch.bfloat16) or st.dtype is not x.dtype:
        return None
    head = st.head
    if head is None or head.export or head.training:
        return None
    if x.shape[2] % st.smax or x.shape[3] % st.smax:
        return None
    if _tm._global_forward_hooks or _tm._global_forward_pre_hooks or any(st.hook_tables):
        return None
    if not st.types_ok:
        return None
    return eng._plan(x, st)


def eligible(model, x):
    """True when this forward may run on the HIP engine (see _plan_if_eligible)."""
    return _plan_if_eligible(model, x) is not None


def try_run(model, x):
    """(z, [x_i]) from the HIP engine, or None when this forward is not eligible: ONE eligibility check and plan lookup
    per forward."""
    plan = _plan_if_eligible(model, x)
    if plan is None:
        return None
    return _ENGINES[model]._run(plan, x)


def run(model, x):
    return _engine_of(model).forward(x)


def run_lane(model, x, lane=0, stream=None):
    """try_run on launch plan number `lane` and on `stream`: forwards of one model issued on different streams need
    different lanes (a plan's activation buffers serve one forward at a time).  Lane 0 is the plan model(x) uses."""
    plan = _plan_if_eligible(model, x)
    if plan is None:
        return None
    eng = _ENGINES[model]
    return eng._run(eng.lane_plan(plan, x, lane), x, stream)


def _engine_of(model):
    eng = _ENGINES.get(model)
    if eng is None:
        eng = Engine(model)
        _ENGINES[model] = eng
    return eng


def set_output_ring(model, n):
    """Serving loops: let every launch plan of `model` write its outputs (z and the raw head maps x_i) into n sets of
    tensors allocated once and used in turn, so that a forward allocates nothing.  The tensors a forward returns are
    then overwritten by the n-th forward after it: consume (or copy) them before that.  n = 0 restores fresh output
    tensors per forward, the default (environment: Y5_OUTPUT_RING)."""
    eng = _engine_of(model)
    eng.output_ring = max(0, int(n))
    for plan in list(eng.plans.values()) + [p for extra in eng.lanes.values() for p in extra]:
        plan.set_ring(eng.output_ring)


class GraphedForward:
    """The forward of `model` on the input buffer `x`, captured once as HIP graphs (one per output set) and replayed:
    one graph launch per forward instead of one launch per kernel, for loops where the host's time per step matters
    (measured on yolov5s at batch 32: 0.07 ms of host time per forward instead of 0.16 - 0.25; the device time does not
    improve, the kernels are long enough to hide their launches, and is 0 - 3 % longer).

    `x` is the graph's input buffer: write every new batch INTO it (x.copy_(batch)) before calling.  A call returns
    (z, [x_i]) like the model, in output set k mod `sets`; a set is overwritten `sets` calls later.  The graphs hold the
    addresses of the weights: after any change to the model (another dtype, fuse(), load_state_dict, pruning) a call
    raises Y5Error and the forward must be captured again."""

    def __init__(self, model, x, sets=2):
        eng = _engine_of(model)
        self._model = weakref.ref(model)
        self.x = x
        set_output_ring(model, sets)
        plan = _plan_if_eligible(model, x)
        if plan is None:
            raise _lib.Y5Error('GraphedForward: this model / input does not run on the HIP engine')
        self.signature = eng.state().signature
        self.plan = plan
        with torch.cuda.device(x.device):
            plan.run(x)  # everything lazy (tuning, function attributes) has happened before the capture
            plan._turn = 0
            cur = torch.cuda.current_stream(x.device)
            side = torch.cuda.Stream(x.device)
            side.wait_stream(cur)
            self.graphs, self.outs = [], []
            with torch.cuda.stream(side):
                for _ in range(max(1, int(sets))):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=side):
                        out = plan.run(x)
                    self.graphs.append(g)
                    self.outs.append(out)
            cur.wait_stream(side)
        self.turn = 0

    def __call__(self):
        eng = _ENGINES.get(self._model())
        if eng is None or self.plan.handle is None or eng.state().signature != self.signature:
            raise _lib.Y5Error('GraphedForward: the model changed since the capture; capture it again')
        k = self.turn
        self.turn = (k + 1) % len(self.graphs)
        self.graphs[k].replay()
        eng.calls += 1
        return self.outs[k]


def engine_of(model):
    """The Engine that has served `model` so far, or None (tests and benchmarks read its counters)."""
    return _ENGINES.get(model)


def calls(model):
    """Number of forwards of `model` that ran as HIP kernels."""
    eng = _ENGINES.get(model)
    return eng.calls if eng is not None else 0


# --------------------------------------------------------------------------------------------
class Act:
    """An NHWC fp16 activation: channel slice [off, off + c_pad) of `buf` (pixel pitch ld)."""

    def __init__(self, buf, off, ld, batch, h, w, segs):
        self.buf, self.off, self.ld, self.batch, self.h, self.w = buf, off, ld, batch, h, w
        self.segs = list(segs)  # [(logical channels, padded channels)]

    @property
    def c_pad(self):
        return sum(p for _, p in self.segs)

    @property
    def c_log(self):
        return sum(c for c, _ in self.segs)

    @property
    def ptr(self):
        """Device address, or None for a shape-only tensor that is never materialised."""
        return None if self.buf is None else self.buf.data_ptr() + 2 * self.off

    def slice(self, off, segs):
        return Act(self.buf, self.off + off, self.ld, self.batch, self.h, self.w, segs)


class Cat2(Act):
    """The channel concatenation [nearest-2x-upsampled a | b] as a tensor that is never built: the 1x1 convolutions that
    read it fetch a at half resolution and b directly (y5_conv_cat2_f16)."""

    def __init__(self, a, b):
        super().__init__(None, 0, 0, b.batch, b.h, b.w, list(a.segs) + list(b.segs))
        self.a, self.b = a, b


class Unsupported(Exception):
    pass


def _conv_geometry(conv):
    """(k, s) of an nn.Conv2d the conv kernel can run, else raises Unsupported."""
    k, s, p = conv.kernel_size, conv.stride, conv.padding
    if conv.groups != 1 or tuple(conv.dilation) != (1, 1) or conv.padding_mode != 'zeros':
        raise Unsupported('grouped/dilated conv')
    if k[0] != k[1] or s[0] != s[1] or k[0] not in (1, 3) or s[0] not in (1, 2):
        raise Unsupported('conv kernel %s stride %s' % (k, s))
    if isinstance(p, str) or tuple(p) != (k[0] // 2, k[0] // 2):
        raise Unsupported('conv padding %s' % (p,))
    return k[0], s[0]


def pack_conv_weight(weights, segs, device, dtype=torch.float16):
    """Row groups [(weight (c, Cin_logical, k, k), padded rows)] -> fp16 [rows256][kpad64];
    K index (r*k + q)*cin_pad + c following the input tensor's channel `segs`.  Returns (packed, kpad, rows)."""
    k = weights[0][0].shape[-1]
    cin_pad = sum(p for _, p in segs)
    cin_log = sum(c for c, _ in segs)
    total_rows = sum(p for _, p in weights)
    rows = (total_rows + 255) // 256 * 256   # whole 256-row weight tiles for the widest kernel variant
    kk = k * k * cin_pad
    kpad = (kk + 63) // 64 * 64
    packed = torch.zeros(rows, kpad, dtype=dtype, device=device)
    r0 = 0
    for weight, rpad in weights:
        cout, cin = weight.shape[:2]
        if cin != cin_log or weight.shape[-1] != k:
            raise Unsupported('conv expects %d input channels, tensor has %d' % (cin, cin_log))
        w = weight.detach().to(dtype).permute(0, 2, 3, 1)  # (Cout, k, k, Cin)
        view = packed[r0:r0 + cout, :kk].view(cout, k, k, cin_pad)
        src = dst = 0
        for c, pc in segs:
            view[..., dst:dst + c] = w[..., src:src + c]
            src += c
            dst += pc
        r0 += rpad
    return packed, kpad, rows


class Plan:
    """A recorded launch plan for one (batch, height, width) plus the buffers it owns."""

    def __init__(self, module, x, block=False):
        """module: a models.yolo.Model, or with block=True a single Conv / Bottleneck / BottleneckCSP / SPP / Focus."""
        self.lib = _lib.lib()
        self.device = x.device
        # element type of every activation and weight of the plan: fp16, or bf16 (then only the kernels compiled for it are
        # recorded: the im2col convolution family, the stand-alone decode, pooling / upsampling; no fused first layer, no
        # folded Upsample + Concat, no chained or Detect-fused convolutions)
        self.dtype = torch.bfloat16 if x.dtype is torch.bfloat16 else torch.float16
        self.bf16 = self.dtype is torch.bfloat16
        self.batch, self.cin, self.h, self.w = x.shape
        self.hwc = input_layout(x)  # interleaved uint8 image: only the fused Focus kernel reads it
        self.keep = []          # tensors the plan's device pointers refer to
        self.labels = []        # one human-readable label per op
        self._defer = None      # list collecting (label, descriptor, input) instead of recording, see _emit_conv
        self._ring, self._turn = [], 0  # sets of output tensors handed out in turn (set_ring); empty: fresh tensors per run
        self.regular_widths = False     # built with channel counts padded to regular widths (pad8)
        self.homes = {}
        handle = C.c_void_p()
        _lib.check(self.lib.y5_plan_create(C.byref(handle)))
        self.handle = handle
        if self.bf16:
            _lib.check(self.lib.y5_plan_set_dtype(handle, _lib.BF16))
        try:
            if block:
                self._build_block(module, x)
            else:
                self._build(module, x)
        except Exception:
            self.close()
            raise

    def close(self):
        if self.handle:
            self.lib.y5_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ allocation helpers
    def _new_act(self, batch, h, w, segs):
        c = sum(p for _, p in segs)
        buf = torch.zeros(batch * h * w * c, dtype=self.dtype, device=self.device)
        self.keep.append(buf)
        return Act(buf, 0, c, batch, h, w, segs)

    def _f32(self, t, n):
        out = torch.zeros(n, dtype=torch.float32, device=self.device)
        out[:t.numel()] = t.detach().float().reshape(-1)
        self.keep.append(out)
        return out

    def _pack_weight(self, weights, segs):
        packed, kpad, rows = pack_conv_weight(weights, segs, self.device, self.dtype)
        self.keep.append(packed)
        return packed, kpad, rows

    # ------------------------------------------------------------------ op emitters
    def _emit_conv(self, label, x, weights, k, s, out, bias=None, scale=None, shift=None, slope=None, res=None, image=None, stem=False):
        """Record one convolution. weights: row groups [(tensor, padded rows)] matching out.segs; out: Act slice to
        write; bias/scale/shift: fp32 vectors of out.c_pad entries (or None); slope: LeakyReLU slope or None.
        image: the NCHW input batch when this is Focus's convolution read straight from it (x then only describes
        the space-to-depth tensor that is never built)."""
        for w, _ in weights:
            if w.dtype is not self.dtype or w.device != self.device:
                raise Unsupported("conv weights must have the input's element type and device")
        if [p for _, p in weights] != [p for _, p in out.segs] or [w.shape[0] for w, _ in weights] != [c for c, _ in out.segs]:
            raise Unsupported('conv output slice mismatch')
        cat2 = None
        if isinstance(x, Cat2):
            if not (k == 1 and s == 1 and image is None and self._defer is None):
                raise Unsupported('a folded Upsample + Concat read by something other than a 1x1 convolution')  # pass 2 of _build rules it out
            cat2 = x
        ho = (x.h + 2 * (k // 2) - k) // s + 1
        wo = (x.w + 2 * (k // 2) - k) // s + 1
        if (ho, wo) != (out.h, out.w):
            raise Unsupported('conv output size mismatch')
        packed, kpad, rows = self._pack_weight(weights, x.segs)
        d = _lib.ConvDesc()
        d.dtype = _lib.BF16 if self.bf16 else _lib.F16
        d.x, d.w, d.y = (x.ptr if image is None else image.data_ptr()), packed.data_ptr(), out.ptr
        cpad = out.c_pad
        if bias is not None:
            d.bias = self._f32(bias, cpad).data_ptr()
        if scale is not None:
            d.scale = self._f32(scale, cpad).data_ptr()
            d.shift = self._f32(shift, cpad).data_ptr()
        if res is not None:
            if (res.c_pad, res.h, res.w) != (cpad, ho, wo):
                raise Unsupported('residual shape mismatch')
            d.res, d.res_ld = res.ptr, res.ld
        d.batch, d.h, d.w_in, d.cin, d.x_ld = x.batch, x.h, x.w, x.c_pad, x.ld
        d.ho, d.wo, d.cout, d.y_ld = ho, wo, cpad, out.ld
        d.ksize, d.stride, d.pad = k, s, k // 2
        d.kpad, d.w_rows = kpad, rows
        if slope is None:
            d.act, d.slope = _lib.ACT_NONE, 0.0
        else:
            if not 0.0 < slope <= 1.0:
                raise Unsupported('LeakyReLU slope %g' % slope)
            d.act, d.slope = _lib.ACT_LEAKY, slope
        if image is not None:
            d.batch, d.h, d.w_in, d.cin, d.x_ld = image.shape[0], image.shape[2], image.shape[3], image.shape[1], 0
            u8 = image.dtype is torch.uint8
            if self._defer is not None:  # the caller records the descriptor itself (_focus_conv2)
                self._defer.append(('%s focus+conv3x3 %d->%d @%dx%d' % (label, x.c_log, out.c_log, ho, wo), d, x))
                return
            if self.hwc:
                if stem:
                    raise Unsupported('an interleaved uint8 image is read by the fused Focus kernel only')
                _lib.check(self.lib.y5_plan_add_focus_conv_u8_hwc(self.handle, C.byref(d), self.hwc))
            else:
                if stem:  # a plain first-layer conv on the image itself (x is the shape-only padded NHWC view of it)
                    add = self.lib.y5_plan_add_stem_conv_u8 if u8 else self.lib.y5_plan_add_stem_conv
                else:
                    add = self.lib.y5_plan_add_focus_conv_u8 if u8 else self.lib.y5_plan_add_focus_conv
                _lib.check(add(self.handle, C.byref(d)))
            kind = ('u8 %s/255+' % ('BGR HWC' if self.hwc == _lib.IN_U8_HWC_BGR else 'HWC')) if self.hwc else 'u8/255+' if u8 else ''
            self.labels.append('%s %s%sconv3x3 %d->%d @%dx%d' % (label, kind, 'image ' if stem else 'focus+', x.c_log, out.c_log, ho, wo))
            return
        if cat2 is not None:
            a, b = cat2.a, cat2.b
            d.x, d.cin, d.x_ld = a.ptr, a.c_pad, a.ld
            d.variant = self._tune_cat2(d, cat2)
            _lib.check(self.lib.y5_plan_add_conv_cat2(self.handle, C.byref(d), b.ptr, b.ld, b.c_pad))
            self.labels.append('%s conv1x1 s1 [up2x %d | %d]->%d @%dx%d c%d' % (label, a.c_log, b.c_log, out.c_log, ho, wo, d.variant))
            return
        text = '%s conv%dx%d s%d %d->%d @%dx%d' % (label, k, k, s, x.c_log, out.c_log, ho, wo)
        if self._defer is not None:  # the caller combines the descriptors itself (_csp: chained convolutions)
            self._defer.append((text, d, x))
            return
        d.variant = self._tune(d, x)
        _lib.check(self.lib.y5_plan_add_conv(self.handle, C.byref(d)))
        self.labels.append('%s v%d' % (text, d.variant))

    def _tune_cat2(self, d, cat):
        """Fastest tile configuration of the two-source 1x1 convolution for this shape (timed, cached per process)."""
        if os.environ.get('Y5_AUTOTUNE', '1') == '0':
            return 0
        a, b = cat.a, cat.b
        key = ('cat2', d.batch, d.ho, d.wo, a.c_pad, a.ld, b.c_pad, b.ld, d.cout, d.y_ld, bool(d.res), bool(d.bias), bool(d.scale))
        stream = _lib.current_stream()

        def launch(v):
            d.variant = v
            return self.lib.y5_conv_cat2_f16(C.byref(d), b.ptr, b.ld, b.c_pad, stream)
        if not _replay_ok(key, launch):
            a.buf.uniform_(-1.0, 1.0)
            b.buf.uniform_(-1.0, 1.0)
            best, best_ms = _fastest(range(1, self.lib.y5_conv_cat2_num_variants() + 1), launch)
            _TUNED[key], _TUNED_MS[key] = best or 0, best_ms
        return _TUNED[key]

    def _emit_chain(self, defs, keep_first=0):
        """Record two deferred convolutions, the second a 1x1 of the first one's output, either as ONE kernel
        (y5_conv_chain_f16; with keep_first = 0 the intermediate tensor, which nothing else reads, is not written) or as
        two, whichever measures faster on the real shapes."""
        (t1, d1, x1), (t2, d2, x2) = defs
        stream = _lib.current_stream()
        tune = os.environ.get('Y5_AUTOTUNE', '1') != '0'
        best, best_ms = 0, float('inf')
        if self.lib.y5_conv_chain_f16(C.byref(d1), C.byref(d2), keep_first, 0, stream) == 0:
            key = ('chain', keep_first, d1.batch, d1.h, d1.w_in, d1.cin, d1.x_ld, d1.cout, d1.ksize, d1.stride, bool(d1.res), d2.cin, d2.cout, d2.y_ld)
            def launch(v):
                return self.lib.y5_conv_chain_f16(C.byref(d1), C.byref(d2), keep_first, v, stream)
            if not _replay_ok(key, launch) and tune:
                x1.buf.uniform_(-1.0, 1.0)
                best, best_ms = _fastest(range(1, self.lib.y5_conv_chain_num_variants() + 1), launch)
                _TUNED[key], _TUNED_MS[key] = best or 0, best_ms
            best, best_ms = _TUNED.get(key, 0), _TUNED_MS.get(key, 0.0 if not tune else float('inf'))
        else:
            best = -1
        # the two-kernel alternative, each with its own fastest variant
        d1.variant = self._tune(d1, x1)
        ms1 = _TUNED_MS.get(self._last_key, float('inf'))
        d2.variant = self._tune(d2, x2)
        ms2 = _TUNED_MS.get(self._last_key, float('inf'))
        if best >= 0 and (not tune or best_ms < ms1 + ms2):
            _lib.check(self.lib.y5_plan_add_conv_chain(self.handle, C.byref(d1), C.byref(d2), keep_first, best))
            self.labels.append('%s -> %s, one kernel v%d' % (t1, t2.split(' conv')[0], best))
            return
        _lib.check(self.lib.y5_plan_add_conv(self.handle, C.byref(d1)))
        self.labels.append('%s v%d' % (t1, d1.variant))
        _lib.check(self.lib.y5_plan_add_conv(self.handle, C.byref(d2)))
        self.labels.append('%s v%d' % (t2, d2.variant))

    def _emit_deferred(self, deferred):
        """Record a convolution that _emit_conv handed back (self._defer) as a kernel of its own."""
        text, d, x = deferred
        d.variant = self._tune(d, x)
        _lib.check(self.lib.y5_plan_add_conv(self.handle, C.byref(d)))
        self.labels.append('%s v%d' % (text, d.variant))

    def _tune(self, d, x):
        """Pick the fastest compiled tile configuration for this convolution by timing every applicable one on
        the real shape (random activations, the layer's own weights).  Results are cached per shape for the
        process; Y5_AUTOTUNE=0 keeps the rule-based choice (variant 0)."""
        key = (d.batch, d.h, d.w_in, d.cin, d.x_ld, d.cout, d.y_ld, d.ksize, d.stride, bool(d.res), bool(d.bias), bool(d.scale), d.dtype)
        self._last_key = key
        if os.environ.get('Y5_AUTOTUNE', '1') == '0':
            return 0
        stream = _lib.current_stream()
        nvar = self.lib.y5_conv_num_variants_bf16() if self.bf16 else self.lib.y5_conv_num_variants()

        def launch(v):
            d.variant = v
            return self.lib.y5_conv_f16(C.byref(d), stream)
        if _replay_ok(key, launch):
            return _TUNED[key]
        x.buf.uniform_(-1.0, 1.0)
        best, best_t = _fastest(range(nvar + 1), launch)
        best = best or 0
        _TUNED[key] = best
        _TUNED_MS[key] = best_t
        return best

    @staticmethod
    def _slope(act):
        if act is None or isinstance(act, nn.Identity):
            return None
        if isinstance(act, nn.LeakyReLU):
            return float(act.negative_slope)
        raise Unsupported('activation %s' % type(act).__name__)

    def _conv(self, label, x, conv, out, bias=None, affine=None, act=None, res=None, image=None, stem=False):
        """One nn.Conv2d with optional bias, folded BatchNorm (scale, shift) and activation module."""
        k, s = _conv_geometry(conv)
        if len(out.segs) != 1:
            raise Unsupported('conv output slice mismatch')
        self._emit_conv(label, x, [(conv.weight, out.segs[0][1])], k, s, out, bias=bias,
                        scale=None if affine is None else affine[0], shift=None if affine is None else affine[1],
                        slope=self._slope(act), res=res, image=image, stem=stem)

    def _conv_block(self, label, m, x, out, res=None, image=None, stem=False):
        """A models.common.Conv (conv + BatchNorm + activation, or its fused form)."""
        if m.bn is None:
            self._conv(label, x, m.conv, out, bias=m.conv.bias, act=m.act, res=res, image=image, stem=stem)
        else:
            if m.conv.bias is not None:
                raise Unsupported('unfused Conv with conv bias')
            self._conv(label, x, m.conv, out, affine=bn_affine(m.bn), act=m.act, res=res, image=image, stem=stem)

    def _as_f16(self, image):
        """The NCHW input as fp16: itself, or for a uint8 input a buffer filled by a normalise op (x.half() / 255)."""
        if self.hwc:
            raise Unsupported('an interleaved uint8 image is read by the fused Focus kernel only')
        if image.dtype is not torch.uint8:
            return image
        out = torch.zeros(image.shape, dtype=torch.float16, device=self.device)
        self.keep.append(out)
        _lib.check(self.lib.y5_plan_add_normalize_u8(self.handle, image.data_ptr(), out.data_ptr(), image.numel()))
        self.labels.append('input u8 -> f16 / 255')
        return out

    def _focus(self, name, m, image, out):
        """models.common.Focus on the plan's input image (NCHW): one fused kernel when its convolution is the usual
        3x3 stride-1 over 3 image channels and at most 96 filters, else space-to-depth packing + convolution."""
        B, cin, H, W = image.shape
        k, s = _conv_geometry(m.conv.conv)
        segs = [(4 * cin, pad8(4 * cin))]
        if cin == 3 and (k, s) == (3, 1) and out.c_pad <= 96 and not self.bf16 and os.environ.get('Y5_FOCUS_FUSE', '1') != '0':
            shape_only = Act(None, 0, segs[0][1], B, H // 2, W // 2, segs)
            self._conv_block(name, m.conv, shape_only, out, image=image)
            return
        image = self._as_f16(image)
        packed = self._new_act(B, H // 2, W // 2, segs)
        _lib.check(self.lib.y5_plan_add_pack_image(self.handle, image.data_ptr(), packed.ptr, B, cin, H, W, 1, packed.ld))
        self.labels.append('%s pack focus' % name)
        self._conv_block(name, m.conv, packed, out)

    def _focus_conv2(self, name0, m0, name1, m1, image, out1):
        """Focus (layer 0) and the 3x3 stride-2 Conv that reads it (layer 1) as ONE kernel (y5_focus_conv2_f16) when the
        pair has the shape that kernel is compiled for and it measures faster than the two kernels; otherwise the two
        layers are recorded as usual.  The tensor between them is only allocated for the two-kernel plan."""
        B, cin, H, W = image.shape
        c0 = m0.conv.conv.out_channels
        segs = [(4 * cin, pad8(4 * cin))]
        shape_only = Act(None, 0, segs[0][1], B, H // 2, W // 2, segs)
        keep_n = len(self.keep)
        mid = self._new_act(B, H // 2, W // 2, [(c0, pad8(c0))])
        self._defer = []
        try:
            self._conv_block(name0, m0.conv, shape_only, mid, image=image)
            self._conv_block(name1, m1, mid, out1)
        finally:
            defs, self._defer = self._defer, None
        (t0, d0, _), (t1, d1, _) = defs
        stream = _lib.current_stream()
        d0.variant = 1  # the software-pipelined order (layer 1 of a tile beside layer 0 of the next) unless the tuner measures otherwise
        fused = self.lib.y5_focus_conv2_f16(C.byref(d0), C.byref(d1), stream) == 0
        d1.variant = self._tune(d1, mid)
        two_ms = _TUNED_MS.get(self._last_key, float('inf'))
        if fused and os.environ.get('Y5_AUTOTUNE', '1') != '0':
            key = ('stem2', B, H, W, out1.ld, bool(d0.bias), bool(d1.bias))
            if key not in _TUNED:  # 0: two kernels, 1 + v: one kernel, variant v
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

                def timed(fn, n=8):
                    fn()
                    e0.record()
                    for _ in range(n):
                        fn()
                    e1.record()
                    e1.synchronize()
                    return e0.elapsed_time(e1) / n

                def one(v):
                    d0.variant = v
                    return timed(lambda: self.lib.y5_focus_conv2_f16(C.byref(d0), C.byref(d1), stream))
                ms = [timed(lambda: self.lib.y5_focus_conv_f16(C.byref(d0), stream)) + two_ms, one(0), one(1)]
                best = min(range(3), key=lambda i: ms[i])
                _TUNED[key], _TUNED_MS[key] = best, ms[best]
            fused = _TUNED[key] > 0
            d0.variant = max(int(_TUNED[key]) - 1, 0)
        if fused:
            del self.keep[keep_n]  # the tensor between the layers: not needed (the packed weights behind it are)
            _lib.check(self.lib.y5_plan_add_focus_conv2(self.handle, C.byref(d0), C.byref(d1)))
            self.labels.append('%s -> %s, one kernel v%d' % (t0, t1, d0.variant))
            return
        _lib.check(self.lib.y5_plan_add_focus_conv(self.handle, C.byref(d0)))
        self.labels.append(t0)
        _lib.check(self.lib.y5_plan_add_conv(self.handle, C.byref(d1)))
        self.labels.append('%s v%d' % (t1, d1.variant))

    def _out_for(self, idx, batch, h, w, c_log):
        """Destination of top-level layer idx: its slice of a later concat buffer, or a fresh buffer."""
        home = self.homes.get(idx)
        segs = [(c_log, pad8(c_log))]
        if home is not None:
            buf_act, off = home
            if (buf_act.h, buf_act.w) == (h, w):
                return buf_act.slice(off, segs)
        return self._new_act(batch, h, w, segs)

    # ------------------------------------------------------------------ graph lowering
    def _build(self, model, x):
        from yolo_amd.models import common, yolo
        layers = list(model.model)
        head = layers[-1]
        B, H, W = self.batch, self.h, self.w

        # ---- pass 1: logical channels / spatial size of every top-level output
        shapes = []  # (c_log, h, w) or None for Detect
        for i, m in enumerate(layers):
            def src(j):
                return (self.cin, H, W) if (i == 0 and j == -1) else shapes[i + j if j < 0 else j]
            f = m.f
            if isinstance(m, common.Focus):
                c, h, w = src(f)
                k, s = _conv_geometry(m.conv.conv)
                shapes.append((m.conv.conv.out_channels, (h // 2 - 1) // s + 1, (w // 2 - 1) // s + 1))
            elif isinstance(m, common.Conv):
                c, h, w = src(f)
                k, s = _conv_geometry(m.conv)
                shapes.append((m.conv.out_channels, (h - 1) // s + 1, (w - 1) // s + 1))
            elif isinstance(m, common.BottleneckCSP):
                c, h, w = src(f)
                shapes.append((m.cv4.conv.out_channels, h, w))
            elif isinstance(m, common.SPP):
                c, h, w = src(f)
                shapes.append((m.cv2.conv.out_channels, h, w))
            elif isinstance(m, common.Bottleneck):
                c, h, w = src(f)
                shapes.append((m.cv2.conv.out_channels, h, w))
            elif isinstance(m, nn.Upsample):
                c, h, w = src(f)
                sf = m.scale_factor
                sf = sf if isinstance(sf, (tuple, list)) else (sf, sf)
                if m.size is not None or tuple(float(v) for v in sf) != (2.0, 2.0) or m.mode not in ('nearest', 'bilinear'):
                    raise Unsupported('upsample %s x%s' % (m.mode, m.scale_factor))
                if m.mode == 'bilinear' and m.align_corners:
                    raise Unsupported('bilinear align_corners=True')
                shapes.append((c, 2 * h, 2 * w))
            elif isinstance(m, common.Concat):
                if m.d != 1:
                    raise Unsupported('concat dim %d' % m.d)
                parts = [src(j) for j in f]
                if len({(p[1], p[2]) for p in parts}) != 1:
                    raise Unsupported('concat of different spatial sizes')
                shapes.append((sum(p[0] for p in parts), parts[0][1], parts[0][2]))
            elif isinstance(m, yolo.Detect):
                shapes.append(None)
            else:
                raise Unsupported('module %s' % type(m).__name__)

        # ---- pass 2: concat buffers and the in-place homes of their operands
        self.homes = {}          # producer layer index -> (concat Act, channel offset)
        self.concat_acts = {}    # concat layer index -> Act
        producer_types = (common.Conv, common.Focus, common.BottleneckCSP, common.SPP, common.Bottleneck, nn.Upsample)
        # nn.Upsample(nearest, 2) -> Concat([up, skip]) read only by 1x1 convolutions: neither tensor is built, the
        # convolutions read the low-resolution tensor and the skip tensor themselves (Cat2)
        consumers = {j: [] for j in range(len(layers))}
        for i, m in enumerate(layers):
            for j in (m.f if isinstance(m.f, (list, tuple)) else [m.f]):
                src_idx = i + j if j < 0 else j
                if src_idx >= 0:
                    consumers[src_idx].append(i)
        self.lazy_cat = {}   # concat layer index -> (upsample layer index, skip layer index)
        self.lazy_up = {}    # upsample layer index -> its source layer index
        if os.environ.get('Y5_CAT2_FUSE', '1') != '0' and not self.bf16:
            for i, m in enumerate(layers):
                if not isinstance(m, common.Concat) or not isinstance(m.f, (list, tuple)) or len(m.f) != 2:
                    continue
                ju, jb = [i + j if j < 0 else j for j in m.f]
                up = layers[ju] if ju >= 0 else None
                if not (isinstance(up, nn.Upsample) and up.mode == 'nearest' and consumers[ju] == [i] and isinstance(up.f, int)):
                    continue
                js = ju + up.f if up.f < 0 else up.f
                if js < 0 or isinstance(layers[js], common.Concat) or isinstance(layers[jb], (common.Concat, nn.Upsample)) or jb == ju:
                    continue
                if shapes[js][0] % 64 != 0 or len(consumers[i]) != 1:
                    continue
                cons = layers[consumers[i][0]]
                if isinstance(cons, common.BottleneckCSP):
                    ok = _conv_geometry(cons.cv1.conv) == (1, 1) and _conv_geometry(cons.cv2) == (1, 1)
                else:
                    ok = isinstance(cons, common.Conv) and _conv_geometry(cons.conv) == (1, 1)
                if ok:
                    self.lazy_cat[i] = (ju, jb)
                    self.lazy_up[ju] = js
        for i, m in enumerate(layers):
            if not isinstance(m, common.Concat) or i in self.lazy_cat:
                continue
            srcs = [i + j if j < 0 else j for j in m.f]
            segs = []
            for j in srcs:
                segs.extend(self.concat_acts[j].segs if j in self.concat_acts else [(shapes[j][0], pad8(shapes[j][0]))])
            _, h, w = shapes[i]
            act = self._new_act(B, h, w, segs)
            self.concat_acts[i] = act
            off = 0
            for j in srcs:
                width = self.concat_acts[j].c_pad if j in self.concat_acts else pad8(shapes[j][0])
                if j not in self.homes and j not in self.concat_acts and isinstance(layers[j], producer_types):
                    self.homes[j] = (act, off)
                off += width

        # ---- pass 3: emit ops
        image = torch.zeros_like(x)  # stands for external slot 0 while recording
        self.keep.append(image)
        _lib.check(self.lib.y5_plan_bind_slot(self.handle, 0, image.data_ptr(), image.numel() * image.element_size()))
        outs = [None] * len(layers)
        # Detect's convolutions of all but the last level only feed the final decode: with Y5_SIDE_STREAM=1 each is
        # recorded right after the layer it reads, on the plan's side stream, so that it runs beside the rest of the
        # head instead of after it.  Off by default: measured at batch 32 it changes the forward time by < 0.5 % (the
        # chip is busy either way), and one stream keeps the plan simpler to reason about.
        # Y5_SIDE_STREAM=2 does the same with the one-kernel levels (convolution + decode): 1.326 -> 1.313-1.323 ms
        # (tools/exp_side_detect.py), also off.
        self.head_bufs = {}
        self.detect_done = set()
        self._decode_desc = None
        early = {}
        side_mode = os.environ.get('Y5_SIDE_STREAM', '0')
        if isinstance(head, yolo.Detect) and side_mode in ('1', '2'):
            srcs = [len(layers) - 1 + j if j < 0 else j for j in (head.f if isinstance(head.f, (list, tuple)) else [head.f])]
            if len(srcs) == head.nl and len(set(srcs)) == len(srcs):
                early = {j: lvl for lvl, j in enumerate(srcs) if j != max(srcs)}
                if side_mode == '2':
                    if self.bf16 or os.environ.get('Y5_DETECT_FUSE', '1') == '0':
                        early = {}
                    else:
                        self._detect_setup(head, [shapes[j][1:] for j in srcs])
        skip = set()
        for i, m in enumerate(layers):
            name = '%d.%s' % (i, type(m).__name__)
            f = m.f

            def get(j):
                return outs[i + j if j < 0 else j]
            if isinstance(m, yolo.Detect):
                self._detect(m, [get(j) for j in f])
                continue
            c_log, h, w = shapes[i]
            if i in skip:
                continue
            if isinstance(m, common.Focus):
                if i != 0 or f != -1:
                    raise Unsupported('Focus must be the first layer')
                nxt = layers[1] if len(layers) > 2 else None
                if (isinstance(nxt, common.Conv) and nxt.f == -1 and image.dtype is torch.float16 and not self.hwc and self.cin == 3
                        and _conv_geometry(m.conv.conv) == (3, 1) and m.conv.conv.out_channels == 32 and _conv_geometry(nxt.conv) == (3, 2)
                        and nxt.conv.out_channels == 64 and nxt.conv.groups == 1 and 0 not in self.homes
                        and not any(0 in [k if k >= 0 else j + 2 + k for k in (l.f if isinstance(l.f, (list, tuple)) else [l.f])] for j, l in enumerate(layers[2:]))
                        and os.environ.get('Y5_STEM2_FUSE', '1') != '0' and os.environ.get('Y5_FOCUS_FUSE', '1') != '0'):
                    c1_log, h1, w1 = shapes[1]
                    out1 = self._out_for(1, B, h1, w1, c1_log)
                    self._focus_conv2(name, m, '1.%s' % type(nxt).__name__, nxt, image, out1)
                    outs[0], outs[1] = None, out1
                    skip.add(1)
                    continue
                out = self._out_for(i, B, h, w, c_log)
                self._focus(name, m, image, out)
            elif isinstance(m, common.Conv):
                if (i == 0 and self.cin == 3 and _conv_geometry(m.conv) == (3, 1) and pad8(c_log) <= 64 and W % 2 == 0
                        and not self.bf16 and os.environ.get('Y5_STEM_FUSE', '1') != '0'):
                    # first-layer 3x3 conv straight from the NCHW (fp16 or uint8) image: no packed copy of the image
                    shape_only = Act(None, 0, 8, B, H, W, [(3, 8)])
                    out = self._out_for(i, B, h, w, c_log)
                    self._conv_block(name, m, shape_only, out, image=image, stem=True)
                    outs[i] = out
                    continue
                if i == 0:
                    xin = self._new_act(B, H, W, [(self.cin, pad8(self.cin))])
                    image = self._as_f16(image)
                    _lib.check(self.lib.y5_plan_add_pack_image(self.handle, image.data_ptr(), xin.ptr, B, self.cin, H, W, 0, xin.ld))
                    self.labels.append('%s pack nchw' % name)
                else:
                    xin = get(f)
                out = self._out_for(i, B, h, w, c_log)
                self._conv_block(name, m, xin, out)
            elif isinstance(m, common.BottleneckCSP):
                out = self._out_for(i, B, h, w, c_log)
                self._csp(name, m, get(f), out)
            elif isinstance(m, common.Bottleneck):
                out = self._out_for(i, B, h, w, c_log)
                self._bottleneck(name, m, get(f), out)
            elif isinstance(m, common.SPP):
                out = self._out_for(i, B, h, w, c_log)
                self._spp(name, m, get(f), out)
            elif isinstance(m, nn.Upsample) and i in self.lazy_up:
                out = get(f)   # stands for its own 2x upsampling until the Concat below wraps it
            elif isinstance(m, common.Concat) and i in self.lazy_cat:
                ju, jb = self.lazy_cat[i]
                out = Cat2(outs[ju], outs[jb])
            elif isinstance(m, nn.Upsample):
                xin = get(f)
                out = self._out_for(i, B, h, w, c_log)
                if len(xin.segs) != 1:
                    out = self._new_act(B, h, w, xin.segs)
                _lib.check(self.lib.y5_plan_add_upsample2x(self.handle, xin.ptr, xin.ld, out.ptr, out.ld, B, xin.h, xin.w,
                                                            xin.c_pad, 0 if m.mode == 'nearest' else 1))
                self.labels.append('%s upsample2x %s' % (name, m.mode))
            elif isinstance(m, common.Concat):
                out = self.concat_acts[i]
                off = 0
                for j in [i + j if j < 0 else j for j in f]:
                    part = outs[j]
                    if not (part.buf is out.buf and part.off == out.off + off):
                        _lib.check(self.lib.y5_plan_add_copy_channels(self.handle, part.ptr, part.ld, out.ptr + 2 * off, out.ld,
                                                                       B * h * w, part.c_pad))
                        self.labels.append('%s copy part %d' % (name, j))
                    off += part.c_pad
            outs[i] = out
            if i in early:
                if side_mode == '2':
                    self._detect_early(head, early[i], out)  # a level that does not fuse is recorded with the rest at the end
                else:
                    self._detect_conv(head, early[i], out, side=True)

    def _build_block(self, m, x):
        """Lower one building block on an NCHW fp16 input (unit tests and debugging)."""
        from yolo_amd.models import common
        B, H, W = self.batch, self.h, self.w
        image = torch.zeros_like(x)
        self.keep.append(image)
        _lib.check(self.lib.y5_plan_bind_slot(self.handle, 0, image.data_ptr(), image.numel() * 2))
        if isinstance(m, common.Focus):
            k, s = _conv_geometry(m.conv.conv)
            c = m.conv.conv.out_channels
            out = self._new_act(B, (H // 2 - 1) // s + 1, (W // 2 - 1) // s + 1, [(c, pad8(c))])
            self._focus('focus', m, image, out)
            self.block_out = out
            return
        xin = self._new_act(B, H, W, [(self.cin, pad8(self.cin))])
        _lib.check(self.lib.y5_plan_add_pack_image(self.handle, image.data_ptr(), xin.ptr, B, self.cin, H, W, 0, xin.ld))
        self.labels.append('pack')
        conv = m.conv if isinstance(m, common.Conv) else m.cv4.conv if isinstance(m, common.BottleneckCSP) else m.cv2.conv
        s = _conv_geometry(conv)[1] if isinstance(m, common.Conv) else 1
        out = self._new_act(B, (xin.h - 1) // s + 1, (xin.w - 1) // s + 1, [(conv.out_channels, pad8(conv.out_channels))])
        if isinstance(m, common.Conv):
            self._conv_block('conv', m, xin, out)
        elif isinstance(m, common.Bottleneck):
            self._bottleneck('bottleneck', m, xin, out)
        elif isinstance(m, common.BottleneckCSP):
            self._csp('csp', m, xin, out)
        elif isinstance(m, common.SPP):
            self._spp('spp', m, xin, out)
        else:
            raise Unsupported('block %s' % type(m).__name__)
        self.block_out = out

    def run_block(self, x):
        """Execute a block plan; returns the NCHW fp16 result."""
        arr = (C.c_void_p * 1)(x.contiguous().data_ptr())
        _lib.check(self.lib.y5_plan_run(self.handle, arr, 1, _lib.current_stream()))
        o = self.block_out
        t = o.buf.view(o.batch, o.h, o.w, o.ld)[..., o.off:o.off + o.c_log]
        return t.permute(0, 3, 1, 2).contiguous()

    def _bottleneck(self, name, m, x, out):
        hidden = self._new_act(x.batch, x.h, x.w, [(m.cv1.conv.out_channels, pad8(m.cv1.conv.out_channels))])
        self._conv_block(name + '.cv1', m.cv1, x, hidden)
        self._conv_block(name + '.cv2', m.cv2, hidden, out, res=x if m.add else None)

    def _csp(self, name, m, x, out):
        """BottleneckCSP (models/common.py:59-62).  cv1 (Conv) and cv2 (plain 1x1) read the same input, so they run
        as ONE convolution whose output rows are [cv1 | cv2]: buffer U = [t | y2].  The BatchNorm + LeakyReLU that
        the reference applies to cat(y1, y2) is split per half: its second half is the epilogue of the cv2 rows, its
        first half the epilogue of cv3, which writes y1 over t's slot once the bottlenecks are done with t - so U
        becomes cat(y1, y2) without a copy and cv4 reads it directly."""
        if m.cv2.bias is not None or m.cv3.bias is not None:
            raise Unsupported('CSP plain convs with bias')
        c1, c2, c3 = m.cv1.conv.out_channels, m.cv2.out_channels, m.cv3.out_channels
        if m.bn.num_features != c3 + c2:
            raise Unsupported('CSP BatchNorm width')
        if self._csp_one_kernel(name, m, x, out):
            return
        self._csp_layers(name, m, x, out)

    def _stage(self, weights, segs, bias=None, affine=None, n=None):
        """y5_csp_stage of one convolution: packed weights + fp32 parameter vectors of n entries."""
        packed, kpad, _ = self._pack_weight(weights, segs)
        st = _lib.CspStage()
        st.w, st.w_ld = packed.data_ptr(), kpad
        if bias is not None:
            st.bias = self._f32(bias, n).data_ptr()
        if affine is not None:
            st.scale, st.shift = self._f32(affine[0], n).data_ptr(), self._f32(affine[1], n).data_ptr()
        return st

    def _conv_stage(self, cm, segs):
        """y5_csp_stage of a models.common.Conv (fused: bias; unfused: folded BatchNorm)."""
        n = cm.conv.out_channels
        if cm.bn is None:
            return self._stage([(cm.conv.weight, n)], segs, bias=cm.conv.bias, n=n)
        if cm.conv.bias is not None:
            raise Unsupported('unfused Conv with conv bias')
        return self._stage([(cm.conv.weight, n)], segs, affine=bn_affine(cm.bn), n=n)

    def _csp_one_kernel(self, name, m, x, out):
        """The whole block as one kernel (y5_csp_f16) when it has one Bottleneck and the channel plan the kernel is
        compiled for, and when that measures faster on this shape than the layer-by-layer plan.  True when recorded."""
        from yolo_amd.models import common
        if self.bf16 or os.environ.get('Y5_CSP_FUSE', '1') == '0' or len(m.m) != 1 or isinstance(x, Cat2) or self._defer is not None:
            return False
        b = m.m[0]
        ch = m.cv1.conv.out_channels
        convs = [m.cv1.conv, m.cv2, b.cv1.conv, m.cv3, m.cv4.conv]
        slopes = {self._slope(a) for a in (m.cv1.act, b.cv1.act, b.cv2.act, m.act, m.cv4.act)}
        if not (type(b) is common.Bottleneck and len(x.segs) == 1 and len(out.segs) == 1
                and (x.c_log, ch, out.c_log) == (64, 32, 64) and x.c_pad == 64 and out.c_pad == 64
                and all(c.out_channels == ch for c in (m.cv2, b.cv1.conv, b.cv2.conv, m.cv3))
                and all(_conv_geometry(c) == (1, 1) and c.groups == 1 for c in convs)
                and _conv_geometry(b.cv2.conv) == (3, 1) and b.cv2.conv.groups == 1
                and x.h % 8 == 0 and x.w % 32 == 0 and len(slopes) == 1 and None not in slopes):
            return False
        for c in convs + [b.cv2.conv]:
            if c.weight.dtype is not self.dtype or c.weight.device != self.device:
                raise Unsupported("conv weights must have the input's element type and device")
        scale, shift = bn_affine(m.bn)
        dev = self.device
        d = _lib.CspDesc()
        d.x, d.y, d.batch, d.h, d.w = x.ptr, out.ptr, x.batch, x.h, x.w
        d.c1, d.x_ld, d.ch, d.c2, d.y_ld = 64, x.ld, ch, 64, out.ld
        d.shortcut, d.slope = int(bool(b.add)), slopes.pop()
        d.cv1 = self._conv_stage(m.cv1, x.segs)
        d.cv2 = self._stage([(m.cv2.weight, ch)], x.segs, affine=(scale[ch:], shift[ch:]), n=ch)
        seg = [(ch, ch)]
        d.m_cv1, d.m_cv2 = self._conv_stage(b.cv1, seg), self._conv_stage(b.cv2, seg)
        d.cv3 = self._stage([(m.cv3.weight, ch)], seg, affine=(scale[:ch], shift[:ch]), n=ch)
        d.cv4 = self._conv_stage(m.cv4, seg * 2)
        stream = _lib.current_stream()
        key = ('csp1', x.batch, x.h, x.w, x.ld, out.ld, bool(b.add), m.cv1.bn is None)
        force = os.environ.get('Y5_CSP_FUSE') == 'force'
        x.buf.uniform_(-1.0, 1.0)
        if self.lib.y5_csp_f16(C.byref(d), stream) != 0:
            # the launcher does not take this block (e.g. a post-op pattern that is neither all bias nor all scale +
            # shift): the layer-by-layer plan does, whatever a cached choice or Y5_CSP_FUSE=force says
            return False
        if os.environ.get('Y5_AUTOTUNE', '1') != '0' and key not in _TUNED and not force:
            # time the kernel against the layer-by-layer plan of the same block, recorded into a scratch plan
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                self.lib.y5_csp_f16(C.byref(d), stream)
            e1.record()
            e1.synchronize()
            one_ms = e0.elapsed_time(e1) / 8
            real, keep_n, label_n = self.handle, len(self.keep), len(self.labels)
            scratch = C.c_void_p()
            _lib.check(self.lib.y5_plan_create(C.byref(scratch)))
            self.handle = scratch
            try:
                self._csp_layers(name, m, x, out)
                arr = (C.c_void_p * 1)(0)
                _lib.check(self.lib.y5_plan_run(scratch, arr, 0, stream))
                e0.record()
                for _ in range(8):
                    self.lib.y5_plan_run(scratch, arr, 0, stream)
                e1.record()
                e1.synchronize()
                layers_ms = e0.elapsed_time(e1) / 8
            finally:
                self.handle = real
                torch.cuda.synchronize()
                self.lib.y5_plan_destroy(scratch)
                del self.keep[keep_n:]
                del self.labels[label_n:]
            _TUNED[key], _TUNED_MS[key] = int(one_ms < layers_ms), min(one_ms, layers_ms)
        if not force and not _TUNED.get(key, 1):
            return False
        _lib.check(self.lib.y5_plan_add_csp(self.handle, C.byref(d)))
        self.labels.append('%s whole block 64->(32)->64 @%dx%d, one kernel' % (name, x.h, x.w))
        return True

    def _csp_layers(self, name, m, x, out):
        """BottleneckCSP layer by layer (see _csp)."""
        head = None
        c1, c2, c3 = m.cv1.conv.out_channels, m.cv2.out_channels, m.cv3.out_channels
        scale, shift = bn_affine(m.bn)
        slope = self._slope(m.act)
        fuse = (pad8(c1) == pad8(c3) and len(m.m) > 0 and self._slope(m.cv1.act) == slope and slope is not None
                and _conv_geometry(m.cv1.conv) == (1, 1) and _conv_geometry(m.cv2) == (1, 1))
        if fuse:
            p1, p2 = pad8(c1), pad8(c2)
            U = self._new_act(x.batch, x.h, x.w, [(c1, p1), (c2, p2)])
            dev = self.device
            one, zero = torch.ones(p1, device=dev), torch.zeros(p1, device=dev)
            if m.cv1.bn is None:   # cv1 already carries its folded bias; identity affine on its rows
                b1 = torch.zeros(p1, device=dev)
                b1[:c1] = m.cv1.conv.bias.detach().float()
                sc1, sh1 = one, zero
            else:
                if m.cv1.conv.bias is not None:
                    raise Unsupported('unfused Conv with conv bias')
                b1 = zero
                s1, h1 = bn_affine(m.cv1.bn)
                sc1, sh1 = one.clone(), zero.clone()
                sc1[:c1], sh1[:c1] = s1, h1
            sc2, sh2 = torch.zeros(p2, device=dev), torch.zeros(p2, device=dev)
            sc2[:c2], sh2[:c2] = scale[c3:], shift[c3:]
            if not isinstance(x, Cat2):
                self._defer = []  # recorded below, alone or chained with the first Bottleneck's cv1
            try:
                self._emit_conv(name + '.cv1+cv2', x, [(m.cv1.conv.weight, p1), (m.cv2.weight, p2)], 1, 1, U,
                                bias=torch.cat((b1, torch.zeros(p2, device=dev))), scale=torch.cat((sc1, sc2)),
                                shift=torch.cat((sh1, sh2)), slope=slope)
            finally:
                head, self._defer = self._defer, None
            t = U.slice(0, [(c1, p1)])
            cat = Act(U.buf, U.off, U.ld, U.batch, U.h, U.w, [(c3, pad8(c3)), (c2, p2)])
        else:
            t = self._new_act(x.batch, x.h, x.w, [(c1, pad8(c1))])
            self._conv_block(name + '.cv1', m.cv1, x, t)
            cat = self._new_act(x.batch, x.h, x.w, [(c3, pad8(c3)), (c2, pad8(c2))])
            self._conv(name + '.cv2', x, m.cv2, cat.slice(pad8(c3), [(c2, pad8(c2))]), affine=(scale[c3:], shift[c3:]), act=m.act)
        bots = list(m.m)
        # The last Bottleneck's cv2 (+ residual) and cv3 as one kernel when both widths fit one tile: the Bottleneck's
        # output, which only cv3 reads, is never written.
        chain = (len(bots) > 0 and not self.bf16 and os.environ.get('Y5_CHAIN_FUSE', '1') != '0' and _conv_geometry(m.cv3) == (1, 1)
                 and max(pad8(bots[-1].cv2.conv.out_channels), pad8(c3)) <= 256)
        # Between Bottlenecks the same kernel serves m[i].cv2 (+ residual) -> m[i+1].cv1 with the first result kept (the
        # next shortcut needs it): the 1x1 then reads its input from LDS instead of HBM.  Measured per shape like the rest.
        link = chain and os.environ.get('Y5_CHAIN_LINK', '1') != '0'
        hidden = None
        if head:
            # cv1 | cv2 and the first Bottleneck's cv1 (a 1x1 of the cv1 half) as one kernel where that measures faster:
            # the half is multiplied again while its tile is still on the CU
            if chain and os.environ.get('Y5_CHAIN_HEAD', '1') != '0' and _conv_geometry(bots[0].cv1.conv) == (1, 1):
                hidden = self._new_act(x.batch, x.h, x.w, [(bots[0].cv1.conv.out_channels, pad8(bots[0].cv1.conv.out_channels))])
                self._defer = []
                try:
                    self._conv_block('%s.m0.cv1' % name, bots[0].cv1, t, hidden)
                finally:
                    defs, self._defer = self._defer, None
                self._emit_chain(head + defs, keep_first=1)
            else:
                self._emit_deferred(head[0])
        for bi, b in enumerate(bots[:-1] if chain else bots):
            nxt = self._new_act(x.batch, x.h, x.w, [(b.cv2.conv.out_channels, pad8(b.cv2.conv.out_channels))])
            if not link:
                if hidden is not None:  # this Bottleneck's cv1 already ran, chained to cv1 | cv2 above
                    self._conv_block('%s.m%d.cv2' % (name, bi), b.cv2, hidden, nxt, res=t if b.add else None)
                    hidden = None
                else:
                    self._bottleneck('%s.m%d' % (name, bi), b, t, nxt)
                t = nxt
                continue
            if hidden is None:
                hidden = self._new_act(x.batch, x.h, x.w, [(b.cv1.conv.out_channels, pad8(b.cv1.conv.out_channels))])
                self._conv_block('%s.m%d.cv1' % (name, bi), b.cv1, t, hidden)
            nb = bots[bi + 1]
            nhidden = self._new_act(x.batch, x.h, x.w, [(nb.cv1.conv.out_channels, pad8(nb.cv1.conv.out_channels))])
            self._defer = []
            try:
                self._conv_block('%s.m%d.cv2' % (name, bi), b.cv2, hidden, nxt, res=t if b.add else None)
                self._conv_block('%s.m%d.cv1' % (name, bi + 1), nb.cv1, nxt, nhidden)
            finally:
                defs, self._defer = self._defer, None
            self._emit_chain(defs, keep_first=1)
            t, hidden = nxt, nhidden
        y1 = cat.slice(0, [(c3, pad8(c3))])
        if chain:
            b, bi = bots[-1], len(bots) - 1
            cm = b.cv2.conv.out_channels
            if hidden is None:
                hidden = self._new_act(x.batch, x.h, x.w, [(b.cv1.conv.out_channels, pad8(b.cv1.conv.out_channels))])
                self._conv_block('%s.m%d.cv1' % (name, bi), b.cv1, t, hidden)
            nxt = self._new_act(x.batch, x.h, x.w, [(cm, pad8(cm))])  # written only if two kernels turn out faster
            self._defer = []
            try:
                self._conv_block('%s.m%d.cv2' % (name, bi), b.cv2, hidden, nxt, res=t if b.add else None)
                self._conv(name + '.cv3', nxt, m.cv3, y1, affine=(scale[:c3], shift[:c3]), act=m.act)
            finally:
                defs, self._defer = self._defer, None
            self._emit_chain(defs)
        else:
            self._conv(name + '.cv3', t, m.cv3, y1, affine=(scale[:c3], shift[:c3]), act=m.act)
        self._conv_block(name + '.cv4', m.cv4, cat, out)

    def _spp(self, name, m, x, out):
        """SPP (models/common.py:74-76). Pool sizes of the form (k, 2k-1, 3k-2, ...) such as (5, 9, 13) are evaluated
        as a cascade of k x k pools, which is exact for max - in one launch when the plane fits in LDS; other size
        lists run each pool directly."""
        ks = [p.kernel_size if isinstance(p.kernel_size, int) else p.kernel_size[0] for p in m.m]
        for p, k in zip(m.m, ks):
            st = p.stride if isinstance(p.stride, int) else p.stride[0]
            pd = p.padding if isinstance(p.padding, int) else p.padding[0]
            if st != 1 or pd != k // 2 or k % 2 != 1 or k > 13 or p.dilation != 1 or p.ceil_mode:
                raise Unsupported('SPP pool k%d s%d p%d' % (k, st, pd))
        c = m.cv1.conv.out_channels
        seg = [(c, pad8(c))]
        cat = self._new_act(x.batch, x.h, x.w, seg * (len(ks) + 1))
        first = cat.slice(0, seg)
        self._conv_block(name + '.cv1', m.cv1, x, first)
        cascade = all(k == n * (ks[0] - 1) + ks[0] for n, k in enumerate(ks))
        if cascade and x.h * x.w * 32 <= 64 * 1024 and len(ks) <= 8:
            dst = cat.slice(pad8(c), seg)
            _lib.check(self.lib.y5_plan_add_maxpool_cascade(self.handle, first.ptr, first.ld, dst.ptr, dst.ld, pad8(c), len(ks),
                                                            x.batch, x.h, x.w, pad8(c), ks[0]))
            self.labels.append('%s maxpool %s' % (name, 'x'.join(str(k) for k in ks)))
            self._conv_block(name + '.cv2', m.cv2, cat, out)
            return
        for n, k in enumerate(ks):
            src = cat.slice(n * pad8(c), seg) if cascade else first
            dst = cat.slice((n + 1) * pad8(c), seg)
            _lib.check(self.lib.y5_plan_add_maxpool(self.handle, src.ptr, src.ld, dst.ptr, dst.ld, x.batch, x.h, x.w, pad8(c),
                                                    ks[0] if cascade else k))
            self.labels.append('%s maxpool%d' % (name, k))
        self._conv_block(name + '.cv2', m.cv2, cat, out)

    def _detect_conv(self, m, lvl, f, side=False):
        """The 1x1 convolution of Detect level lvl on feature map f (models/yolo.py:33) into its own NHWC buffer."""
        conv = m.m[lvl]
        if conv.out_channels != m.na * m.no:
            raise Unsupported('Detect conv width')
        headbuf = self._new_act(self.batch, f.h, f.w, [(m.na * m.no, pad8(m.na * m.no))])
        self._conv('detect.m%d' % lvl, f, conv, headbuf, bias=conv.bias)
        if side:
            _lib.check(self.lib.y5_plan_mark_last(self.handle, _lib.PLAN_SIDE))
            self.labels[-1] += ' (side stream)'
        self.head_bufs[lvl] = headbuf

    def _detect_fused(self, m, lvl, f, dd):
        """Descriptors of Detect level lvl as one kernel (y5_detect_f16: 1x1 conv + decode, models/yolo.py:32-46), with
        the fastest tile configuration for the shape; None when the kernel does not take the level."""
        conv = m.m[lvl]
        cout = m.na * m.no
        if conv.out_channels != cout or pad8(cout) > 256 or conv.bias is None or _conv_geometry(conv) != (1, 1):
            return None
        packed, kpad, rows = self._pack_weight([(conv.weight, pad8(cout))], f.segs)
        cd = _lib.ConvDesc()
        cd.x, cd.w = f.ptr, packed.data_ptr()
        cd.bias = self._f32(conv.bias, pad8(cout)).data_ptr()
        cd.batch, cd.h, cd.w_in, cd.cin, cd.x_ld = f.batch, f.h, f.w, f.c_pad, f.ld
        cd.ho, cd.wo, cd.cout, cd.y_ld = f.h, f.w, pad8(cout), pad8(cout)
        cd.ksize, cd.stride, cd.pad, cd.kpad, cd.w_rows = 1, 1, 0, kpad, rows
        cd.act, cd.slope = _lib.ACT_NONE, 0.0
        stream = _lib.current_stream()
        if self.lib.y5_detect_f16(C.byref(cd), C.byref(dd), 0, stream) != 0:
            return None
        best, best_t = 0, float('inf')
        key = ('detect', cd.batch, cd.h, cd.w_in, cd.cin, cd.x_ld, cd.cout, dd.na, dd.no)
        if key in _TUNED:
            best = _TUNED[key]
        elif os.environ.get('Y5_AUTOTUNE', '1') != '0':
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for v in range(1, self.lib.y5_detect_num_variants() + 1):
                if self.lib.y5_detect_f16(C.byref(cd), C.byref(dd), v, stream) != 0:
                    continue
                e0.record()
                for _ in range(4):
                    self.lib.y5_detect_f16(C.byref(cd), C.byref(dd), v, stream)
                e1.record()
                e1.synchronize()
                if e0.elapsed_time(e1) < best_t:
                    best, best_t = v, e0.elapsed_time(e1)
            _TUNED[key], _TUNED_MS[key] = best, best_t / 4
        label = 'detect.m%d conv1x1+decode %d->%d @%dx%d v%d' % (lvl, f.c_log, cout, f.h, f.w, best)
        return label, cd, dd, best

    def _detect_setup(self, m, hw):
        """Output tensors of Detect (models/yolo.py:24-46) for feature maps of sizes hw = [(h, w), ...], bound as the
        plan's external slots 1, 2, ...; returns the function that fills a level's decode descriptor."""
        B = self.batch
        na, no = m.na, m.no
        if na > _lib.Y5_MAX_ANCHORS:
            raise Unsupported('more than %d anchors' % _lib.Y5_MAX_ANCHORS)
        rows = [na * h * w for h, w in hw]
        self.z_shape = (B, sum(rows), no)
        self.x_shapes = [(B, na, h, w, no) for h, w in hw]
        z = torch.zeros(self.z_shape, dtype=self.dtype, device=self.device)
        xs = [torch.zeros(s, dtype=self.dtype, device=self.device) for s in self.x_shapes]
        self.keep += [z] + xs
        _lib.check(self.lib.y5_plan_bind_slot(self.handle, 1, z.data_ptr(), z.numel() * 2))
        for i, t in enumerate(xs):
            _lib.check(self.lib.y5_plan_bind_slot(self.handle, 2 + i, t.data_ptr(), t.numel() * 2))
        anchor_grid = m.anchor_grid.detach().float().cpu().view(m.nl, na, 2)  # element-type values (the buffer follows .half() / .bfloat16())

        def decode_desc(i, f):
            d = _lib.DecodeDesc()
            d.dtype = _lib.BF16 if self.bf16 else _lib.F16
            d.x_out, d.z_out = xs[i].data_ptr(), z.data_ptr()
            d.batch, d.ny, d.nx, d.na, d.no = B, f.h, f.w, na, no
            d.z_rows_total, d.z_row0 = self.z_shape[1], sum(rows[:i])
            d.stride = float(m.stride[i])
            for a in range(na):
                d.anchor_wh[2 * a] = float(anchor_grid[i, a, 0])
                d.anchor_wh[2 * a + 1] = float(anchor_grid[i, a, 1])
            return d
        self._decode_desc = decode_desc
        return decode_desc

    def _detect_early(self, m, lvl, f):
        """Detect level lvl as one kernel on the plan's side stream, recorded right after the layer that produces its
        feature map f; False when the one-kernel form does not take the level (nothing recorded then)."""
        op = self._detect_fused(m, lvl, f, self._decode_desc(lvl, f))
        if op is None:
            return False
        label, cd, dd, variant = op
        _lib.check(self.lib.y5_plan_add_detect(self.handle, C.byref(cd), C.byref(dd), variant))
        _lib.check(self.lib.y5_plan_mark_last(self.handle, _lib.PLAN_SIDE))
        self.labels.append(label + ' (side stream)')
        self.detect_done.add(lvl)
        return True

    def _detect(self, m, feats):
        na, no = m.na, m.no
        rows = [na * f.h * f.w for f in feats]
        decode_desc = getattr(self, '_decode_desc', None) or self._detect_setup(m, [(f.h, f.w) for f in feats])
        row0 = 0

        # Every level as ONE kernel (head conv with the decode as its epilogue) when all of them qualify and none was
        # already recorded on the side stream; otherwise head convs + one decode launch.
        if not self.head_bufs and not self.bf16 and os.environ.get('Y5_DETECT_FUSE', '1') != '0':
            ops = []
            todo = [i for i in range(len(feats)) if i not in self.detect_done]
            for i in todo:
                op = self._detect_fused(m, i, feats[i], decode_desc(i, feats[i]))
                if op is None:
                    break
                ops.append(op)
            if len(ops) == len(todo):
                for label, cd, dd, variant in ops:
                    _lib.check(self.lib.y5_plan_add_detect(self.handle, C.byref(cd), C.byref(dd), variant))
                    self.labels.append(label)
                if self.detect_done:
                    _lib.check(self.lib.y5_plan_mark_last(self.handle, _lib.PLAN_JOIN))
                return
        if self.detect_done:
            raise Unsupported('Detect levels recorded early but the rest does not fuse')
        fused = len(feats) <= _lib.Y5_MAX_LEVELS and os.environ.get('Y5_DECODE_FUSE', '1') != '0'
        levels = (_lib.DecodeDesc * len(feats))()
        for i, f in enumerate(feats):
            if i not in self.head_bufs:
                self._detect_conv(m, i, f)
            headbuf = self.head_bufs[i]
            d = decode_desc(i, f)
            d.head, d.head_ld = headbuf.ptr, headbuf.ld
            if fused:
                C.memmove(C.byref(levels[i]), C.byref(d), C.sizeof(_lib.DecodeDesc))
            else:
                _lib.check(self.lib.y5_plan_add_decode(self.handle, C.byref(d)))
                _lib.check(self.lib.y5_plan_mark_last(self.handle, _lib.PLAN_JOIN))
                self.labels.append('detect.decode%d @%dx%d' % (i, f.h, f.w))
            row0 += rows[i]
        if fused:  # every level's head conv is recorded by now: one launch decodes them all
            _lib.check(self.lib.y5_plan_add_decode_multi(self.handle, levels, len(feats)))
            _lib.check(self.lib.y5_plan_mark_last(self.handle, _lib.PLAN_JOIN))
            self.labels.append('detect.decode all %d levels' % len(feats))

    # ------------------------------------------------------------------ execution
    def _ext(self, x, z, xs):
        arr = (C.c_void_p * (2 + len(xs)))()
        arr[0], arr[1] = x.data_ptr(), z.data_ptr()
        for i, t in enumerate(xs):
            arr[2 + i] = t.data_ptr()
        return arr

    def set_ring(self, n):
        """Keep n sets of output tensors (z and every x_i) and hand them out in turn instead of allocating per forward;
        n = 0 goes back to fresh tensors.  With a ring the results of forward k are overwritten by forward k + n."""
        self._ring, self._turn = [], 0
        for _ in range(max(0, int(n))):
            z = torch.empty(self.z_shape, dtype=self.dtype, device=self.device)
            xs = [torch.empty(s, dtype=self.dtype, device=self.device) for s in self.x_shapes]
            self._ring.append((z, xs, self._ext(z, z, xs)))

    def run(self, x, stream=None):
        """Issue the forward on `stream` (a torch.cuda.Stream; default: the current one)."""
        ring = self._ring
        if ring:
            z, xs, ext = ring[self._turn]
            self._turn = (self._turn + 1) % len(ring)
            ext[0] = x.data_ptr()
        else:
            z = torch.empty(self.z_shape, dtype=self.dtype, device=self.device)
            xs = [torch.empty(s, dtype=self.dtype, device=self.device) for s in self.x_shapes]
            ext = self._ext(x, z, xs)
        _lib.check(self.lib.y5_plan_run(self.handle, ext, len(ext), _lib.current_stream() if stream is None else C.c_void_p(stream.cuda_stream)))
        return z, xs

    def profile(self, x):
        """Per-op device milliseconds (hipEvent pairs on the current stream) with kind/FLOPs/bytes/labels."""
        z = torch.empty(self.z_shape, dtype=self.dtype, device=self.device)
        xs = [torch.empty(s, dtype=self.dtype, device=self.device) for s in self.x_shapes]
        ext = self._ext(x, z, xs)
        n = self.lib.y5_plan_num_ops(self.handle)
        ms = (C.c_float * n)()
        _lib.check(self.lib.y5_plan_profile(self.handle, ext, len(ext), _lib.current_stream(), ms, n))
        rows = []
        for i in range(n):
            kind, flops, nbytes = C.c_int(), C.c_double(), C.c_double()
            _lib.check(self.lib.y5_plan_op_info(self.handle, i, C.byref(kind), C.byref(flops), C.byref(nbytes)))
            rows.append({'op': i, 'kind': _lib.OP_NAMES.get(kind.value, '?'), 'label': self.labels[i], 'ms': float(ms[i]),
                         'flops': flops.value, 'bytes': nbytes.value})
        return rows


def run_block(module, x):
    """Run one building block (Conv, Focus, Bottleneck, BottleneckCSP, SPP) through the HIP kernels.
    x: (B, C, H, W) fp16 GPU tensor; module: fp16, eval mode, on the same device."""
    with torch.cuda.device(x.device):
        plan = Plan(module, x, block=True)
        try:
            return plan.run_block(x)
        finally:
            torch.cuda.synchronize()
            plan.close()


class _State:
    """What a full walk of the model found, kept until something changes: the tensor list with its version counters and
    addresses (the cheap per-forward comparison), the hook tables of every sub-module, and the facts eligibility needs."""
    __slots__ = ('epoch', 'tensors', 'versions', 'ptrs', 'hook_tables', 'dtype', 'head', 'stride', 'stride_version', 'smax',
                 'types_ok', 'signature')


class Engine:
    """Per-model cache of launch plans keyed by input shape and by the identity/version of every parameter."""

    def __init__(self, model):
        self._model = weakref.ref(model)  # the registry maps model -> Engine weakly; a strong reference here would pin both
        self.plans = collections.OrderedDict()
        self.signature = None
        self.unsupported = None   # (signature, reason) of the last failed lowering
        self.calls = 0            # forwards executed by HIP kernels (read by tests)
        self.walks = 0            # full walks of the model (tests: the steady state does none)
        self.output_ring = int(os.environ.get('Y5_OUTPUT_RING', '0'))  # sets of output tensors each plan reuses; 0 = fresh tensors
        self.lanes = {}           # plan key -> further plans for the same input (lane_plan)
        self._state = None

    @property
    def model(self):
        return self._model()

    def _walk(self):
        """Full walk of the model: O(modules + tensors).  Runs on the first forward and after anything was assigned to a
        module anywhere in the process, a tensor was written in place or moved (version counter / address), or Detect's
        stride attribute was replaced."""
        from yolo_amd.models.yolo import Detect
        model = self.model
        st = _State()
        st.epoch = _REG_EPOCH[0]
        st.tensors = list(model.parameters()) + list(model.buffers())
        st.versions = [t._version for t in st.tensors]
        st.ptrs = [t.data_ptr() for t in st.tensors]
        mods = list(model.modules())
        st.hook_tables = [m._forward_hooks for m in mods] + [m._forward_pre_hooks for m in mods]
        st.dtype = st.tensors[0].dtype if st.tensors else None
        head = model.model[-1]
        st.head = head if isinstance(head, Detect) else None
        st.stride = getattr(head, 'stride', None)
        st.stride_version = st.stride._version if isinstance(st.stride, torch.Tensor) else None
        st.smax = int(max(st.stride)) if st.stride is not None else 32
        st.types_ok = all(isinstance(m, _supported_types()) for m in model.model)
        sig = [(id(t), p, v, t.dtype) for t, p, v in zip(st.tensors, st.ptrs, st.versions)]
        sig.append(tuple(float(s) for s in st.stride) if st.stride is not None else None)
        st.signature = hash(tuple(sig))
        self.walks += 1
        return st

    def state(self):
        """The model's state, from the cache when nothing changed since the last walk."""
        st = self._state
        if st is not None and st.epoch == _REG_EPOCH[0]:
            ts = st.tensors
            stride = getattr(self.model.model[-1], 'stride', None)
            if (stride is st.stride and (st.stride_version is None or stride._version == st.stride_version)
                    and list(map(_VERSION_OF, ts)) == st.versions and list(map(_ADDRESS_OF, ts)) == st.ptrs):
                return st
        st = self._state = self._walk()
        if st.signature != self.signature:
            if self.lanes:  # forwards of the old plans may be in flight on other streams than the current one
                torch.cuda.synchronize()
            for p in self.plans.values():
                p.close()
            self.plans.clear()
            self._drop_lanes()
            self.signature = st.signature
        return st

    def _signature(self):
        return self.state().signature

    def _plan(self, x, st=None):
        sig = (st or self.state()).signature
        key = (tuple(x.shape), x.device.index, x.dtype, input_layout(x))
        plan = self.plans.get(key)
        if plan is None:
            if self.unsupported is not None and self.unsupported[0] == (sig, key):
                return None
            try:
                with torch.cuda.device(x.device):
                    plan = Plan(self.model, x)
                    if os.environ.get('Y5_PAD_REGULAR', '1') != '0' and not plan.bf16 and _irregular_widths(self.model):
                        plan = self._faster_padding(plan, x)
            except Unsupported as e:
                self.unsupported = ((sig, key), str(e))
                return None
            plan.set_ring(self.output_ring)
            self.plans[key] = plan
            while len(self.plans) > max(1, MAX_PLANS):  # a rectangular-batch loader feeds many shapes: bound the buffers held
                old_key, old = self.plans.popitem(last=False)
                torch.cuda.synchronize(x.device)  # launches of the evicted plan may still be in flight on any stream
                old.close()
                self._drop_lanes(old_key)
        else:
            self.plans.move_to_end(key)
        return plan

    def supported(self, x):
        st = self.state()
        return st.types_ok and self._plan(x, st) is not None

    def _run(self, plan, x, stream=None):
        if not plan.hwc:
            x = x.contiguous()
        with torch.cuda.device(x.device):
            z, xs = plan.run(x, stream)
        self.calls += 1
        return z, xs

    def _faster_padding(self, plan, x):
        """A model with irregular channel counts: build its plan a second time with the widths padded to regular ones
        (pad8) and keep whichever of the two runs the forward faster; lanes built later follow the choice."""
        def ms(p):
            xs = x if p.hwc else x.contiguous()
            for _ in range(2):
                p.run(xs)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                p.run(xs)
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) / 8
        _PAD_REGULAR[0] = True
        try:
            wide = Plan(self.model, x)
        except Unsupported:
            return plan
        finally:
            _PAD_REGULAR[0] = False
        wide.regular_widths = True
        a, b = ms(plan), ms(wide)
        keep, drop = (wide, plan) if b < a or os.environ.get('Y5_PAD_REGULAR') == 'force' else (plan, wide)
        torch.cuda.synchronize()
        drop.close()
        return keep

    def lane_plan(self, plan, x, lane):
        """Launch plan number `lane` for the input that `plan` (lane 0, the one model(x) uses) was built for.  Lanes are
        independent copies of the plan - own activation buffers, own outputs - so that forwards issued on different streams
        can be in flight together.  They live and die with the model's state like every plan."""
        if lane == 0:
            return plan
        key = next(k for k, p in self.plans.items() if p is plan)
        extra = self.lanes.setdefault(key, [])
        while len(extra) < lane:
            with torch.cuda.device(x.device):
                _PAD_REGULAR[0] = plan.regular_widths
                try:
                    p = Plan(self.model, x)
                finally:
                    _PAD_REGULAR[0] = False
                p.regular_widths = plan.regular_widths
            p.set_ring(self.output_ring)
            extra.append(p)
        return extra[lane - 1]

    def _drop_lanes(self, key=None):
        for k in ([key] if key is not None else list(self.lanes)):
            for p in self.lanes.pop(k, []):
                p.close()

    def forward(self, x):
        plan = self._plan(x)
        if plan is None:
            raise _lib.Y5Error('model not supported by the HIP engine: %s' % (self.unsupported or ('', ''))[1])
        return self._run(plan, x)

    def profile(self, x):
        plan = self._plan(x)
        with torch.cuda.device(x.device):
            return plan.profile(x.contiguous())
