"""torch.autograd.Function wrappers over the C ABI of libdsrl_hip.so (include/dsrl_hip.h).

Tensors keep the reference's logical NCHW shapes; physically they are torch.channels_last ("pixel-major"
[P][ld] for the kernels), which the wrappers enforce with at most one strided copy at the boundary.  Every op
raises if its tensors are not on a HIP device: there is no CPU / stock-ATen fallback in this package.
"""
import ctypes
import os
import weakref

import torch

from . import _lib
from ._lib import call, query, DsrlHipError

CL = torch.channels_last


# ------------------------------------------------------------------------------------------------ helpers
_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_raw_device = getattr(torch._C, '_cuda_getDevice', None)


def _stream():
    """hipStream_t of torch's current stream on the current device (what `with torch.cuda.stream(...)` selects).  The raw accessors
    cost ~0.3 us; torch.cuda.current_stream() builds a Stream object and re-checks device availability (~4 us) on each of the ~250
    calls of a step."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


# Weight-gradient kernels run on a side HIP stream so that they overlap with the data-gradient kernel of the same layer (both
# only read dy): the small backbone layers do not fill 256 CUs on their own. ddp.FlatParams joins the stream before it
# reduces / applies the gradients; without an arena the caller's stream waits right away.
_side = {}
overlap_wgrad = os.environ.get('DSRL_OVERLAP_WGRAD', '1') != '0'     # measured: +10-13 % step throughput on one MI355X
# building dgrad's transposed filters during the forward pass on the side stream: measured SLOWER (two-stream allocator / event
# overhead on the host outweighs the 0.5 ms of transposes it hides) - kept selectable
pretranspose_filters = os.environ.get('DSRL_PRETRANSPOSE', '0') != '0'


CONV_PRECISION_MODES = {'fp32': 0, 'bf16x3': 1, 'bf16x6': 2, 'mixed': 3, 'f16x3': 4, 'f16x1': 5}
DEFAULT_CONV_PRECISION = 4


def set_conv_precision(mode):
    """Arithmetic of the conv kernels (include/dsrl_hip.h: dsrl_conv_precision): 'fp32' (exact fp32 MFMA products), 'bf16x3',
    'bf16x6' (fp32-equivalent), 'mixed' (forward bf16x6, backward bf16x3), 'f16x3' (two fp16 terms of the per-tensor scaled
    operands: fp32-equivalent at half the matrix work of bf16x6; the default since round 3) or 'f16x1' (ONE fp16 term, one MFMA per
    product, fp32 accumulation: apex O1 / O2's arithmetic, the reduced-precision mode); None follows DSRL_CONV_PRECISION.
    Returns the previous setting as the library reported it (an int, -1 = environment)."""
    code = -1 if mode is None else (CONV_PRECISION_MODES[mode] if isinstance(mode, str) else int(mode))
    if not -1 <= code <= 5:
        raise ValueError(f'conv precision mode {mode!r}')
    _mode_cache.clear()
    _query_cache.clear()        # workspace sizes and partial counts follow the arithmetic's tile / pixel-range plans
    return int(_lib.load().dsrl_conv_precision(code))


_mode_cache = {}


def _conv_precision_code():
    code = _mode_cache.get('code')
    if code is None:
        prev = int(_lib.load().dsrl_conv_precision(-2))          # out-of-range argument: query only
        code = prev if prev >= 0 else int(os.environ.get('DSRL_CONV_PRECISION', str(DEFAULT_CONV_PRECISION)))
        code = _mode_cache['code'] = min(max(code, 0), 5)
    return code


def get_conv_precision():
    """Name of the mode the conv kernels currently run in."""
    code = _conv_precision_code()
    return [k for k, v in CONV_PRECISION_MODES.items() if v == code][0]


# ------------------------------------------------------------------------------------------------ operand magnitudes (f16x3 arithmetic)
# The f16x3 conv kernels scale every operand tensor by a power of two taken from max |x| of the whole tensor (include/dsrl_hip.h:
# dsrl_amax).  A magnitude is one amax record of 256 uint32 device words ("slot"); a tensor that has one carries it as the attribute `_dsrl_amax`, left by
# the kernel that wrote the tensor (BatchNorm apply / backward, the batched filter transpose) or by a dsrl_amax launch here, so that the
# forward conv and the weight gradient (x) and the data and weight gradients (dy) share one measurement.  Slots come from an arena that
# ddp.FlatParams.zero_grad() rewinds and zeroes at the start of every training step (one memset; static addresses under graph capture).
_AMAX_SLOTS, AMAX_WORDS = 2048, 256          # records per arena (a stage-3 step uses ~700); include/dsrl_hip.h: DSRL_AMAX_WORDS
# Two kinds of arenas per device.  The STEP arena belongs to training steps: ddp.FlatParams.zero_grad() rewinds and zeroes it (amax_begin_step),
# sgd_step() closes it (amax_end_step); a captured hipGraph bakes in its addresses (the zero fill, ~700 records the kernels max into and read), so a
# capture PINS it (amax_pin: the graph holds the tensor, and a pinned arena is never replaced).  Everything outside an open step - evaluation
# forwards between epochs, tools, tests - draws from a LOOSE arena that is simply replaced when it is full (views keep the old one alive for the
# tensors that still carry its records).  Round 3 had one arena for both: ten validation batches exhausted it, the replacement freed the memory a
# live graph kept writing to.
_amax_arena = {}            # device -> [int32 tensor, next free slot, generation, pinned]
_amax_loose = {}            # device -> [int32 tensor, next free slot]
_amax_open = set()          # devices whose step arena is open


def _new_arena(device):
    return torch.zeros(_AMAX_SLOTS * AMAX_WORDS, dtype=torch.int32, device=device)


def amax_begin_step(device):
    """Rewinds the step arena of `device` and zeroes it (stream-ordered): every record handed out of it before belongs to a finished step - the
    generation counter invalidates the `_dsrl_amax` attributes that still point into it (amax_for re-measures such tensors)."""
    ar = _amax_arena.get(device)
    if ar is None:
        ar = _amax_arena[device] = [_new_arena(device), 0, 1, False]
    else:
        ar[0].zero_()
        ar[1] = 0
        ar[2] += 1
    _amax_open.add(device)
    return ar


def amax_end_step(device):
    """The training step of `device` is over (its optimiser update has been enqueued, or it was abandoned): records asked for from here on come from
    the loose arena, and the fp16 planes cached on this step's operand tensors are released."""
    _amax_open.discard(device)
    drop_planes()


def amax_pin(device):
    """The step arena of `device`, marked as referenced by a captured graph (the caller keeps the tensor with the graph): never replaced from now on."""
    ar = _amax_arena.get(device)
    if ar is None:
        ar = amax_begin_step(device)
        amax_end_step(device)
    ar[3] = True
    return ar[0]


def amax_slot(device):
    """A zeroed amax record.  The returned view carries `_dsrl_gen`: the generation of the step arena it came from (-1: loose arena, never reused)."""
    ar = _amax_arena.get(device)
    if ar is not None and device in _amax_open:
        if ar[1] >= _AMAX_SLOTS:
            if ar[3]:
                raise DsrlHipError(f'amax arena exhausted inside a training step whose graph is captured ({_AMAX_SLOTS} records): raise functional._AMAX_SLOTS')
            ar[0], ar[1] = _new_arena(device), 0          # an eager step that needs more: views keep the old tensor alive for its kernels
        i = ar[1]
        ar[1] = i + 1
        slot = ar[0][i * AMAX_WORDS:(i + 1) * AMAX_WORDS]
        slot._dsrl_gen = ar[2]
        return slot
    lo = _amax_loose.get(device)
    if lo is None or lo[1] >= _AMAX_SLOTS:
        lo = _amax_loose[device] = [_new_arena(device), 0]
    i = lo[1]
    lo[1] = i + 1
    slot = lo[0][i * AMAX_WORDS:(i + 1) * AMAX_WORDS]
    slot._dsrl_gen = -1
    return slot


def f16_mode():
    """The conv kernels scale their operands by per-tensor powers of two (amax records): 'f16x3' and 'f16x1'."""
    return _conv_precision_code() >= 4


def set_amax(t, slot):
    """Attaches the amax record `slot` to tensor `t` together with what makes it stale: the tensor's version counter (an in-place write) and the
    generation of the step arena the record lives in (amax_begin_step zeroes and re-issues those records)."""
    try:
        t._dsrl_amax = slot
        t._dsrl_amax_meta = (t._version, getattr(slot, '_dsrl_gen', -1))
    except Exception:           # noqa: BLE001
        pass


def carried_amax(t):
    """The record `t` carries if it still describes the tensor's values, else None."""
    slot = getattr(t, '_dsrl_amax', None)
    if slot is None:
        return None
    meta = getattr(t, '_dsrl_amax_meta', None)
    if meta is None or meta[0] != t._version:
        return None
    if meta[1] != -1:
        ar = _amax_arena.get(slot.device)
        if ar is None or ar[2] != meta[1]:
            return None
    return slot


def amax_for(t, data=None, ld=None):
    """The magnitude record of pixel-major tensor `t` (N,C,H,W): the one it carries (left by its producer or by an earlier measurement, and still
    valid: carried_amax), or a fresh measurement that it then carries.  `data`/`ld`: the pixel-major copy of `t` the kernels read, if the caller has it."""
    slot = carried_amax(t)
    if slot is not None:
        return slot
    if data is None:
        data, ld = pm(t)
    N, Cc, H, W = data.shape
    slot = amax_slot(data.device)
    call('dsrl_amax', data.data_ptr(), ld, N * H * W, Cc, slot.data_ptr(), _stream())
    set_amax(t, slot)
    return slot


def set_bn_fused_max_blocks(n):
    """Block budget of the single-kernel BatchNorm (include/dsrl_hip.h: dsrl_bn_fused_max_blocks): 0 = off, 128, 256; None = follow the
    environment. Its device-wide barrier needs all blocks resident at once, i.e. no other process on the GPU."""
    return int(_lib.load().dsrl_bn_fused_max_blocks(-1 if n is None else int(n)))


def bn_fused_barrier_timeouts():
    """Blocks of fused BN launches that gave up waiting at the barrier so far (their outputs were poisoned with NaN). Synchronises."""
    import ctypes
    n = ctypes.c_int64(0)
    _lib.check(_lib.load().dsrl_bn_fused_barrier_timeouts(ctypes.byref(n)), 'dsrl_bn_fused_barrier_timeouts')
    return int(n.value)


def side_stream(device):
    st = _side.get(device)
    if st is None:
        st = _side[device] = torch.cuda.Stream(device=device)
    return st


def join_side_streams():
    if not _side:                   # no side stream was ever made (also: a host-only schedule test on CPU)
        return
    cur = torch.cuda.current_stream()
    for st in _side.values():
        if st.device == cur.device:
            cur.wait_stream(st)


# Deferred weight gradients: while a WgradQueue is open (ddp.FlatParams opens one per training step), a conv's backward only records
# its weight-gradient problem (x, dy, arena slot); flush() launches all of them as a few grouped grids (dsrl_conv2d_wgrad_group_*).
# Nothing but the optimiser reads a weight gradient, the small layers do not fill the chip on their own, and a pass-wide grid needs
# neither the side stream nor per-layer pixel splits.  Off: DSRL_WGRAD_GROUP=0 (per-layer launches, overlapped on the side stream).
group_wgrad = os.environ.get('DSRL_WGRAD_GROUP', '1') != '0'
graph_keepalive = None          # a list while a hipGraph capture is in progress: host buffers the captured copies read on every replay
capture_host, capture_host_off = None, 0      # pinned arena for host tables written during a capture (allocated before it starts)


class WgradQueue:
    def __init__(self):
        self.items = []           # (x, ldx, dy, lddy, dw tensor, shp, on_written)

    def add(self, x, ldx, dy, lddy, dw, shp, on_written=None, x_amax=None, dy_amax=None):
        self.items.append((x, ldx, dy, lddy, dw, shp, on_written, x_amax, dy_amax))

    def flush(self):
        items, self.items = self.items, []
        if not items:
            return
        n = len(items)
        probs = (_lib.WgradProblem * n)()
        for q, (x, ldx, dy, lddy, dw, shp, _, xa, dya) in zip(probs, items):
            N, H, W, Cc, K, R, S, stride, pad, dil = shp
            q.x, q.dy, q.dw = x.data_ptr(), dy.data_ptr(), dw.data_ptr()
            q.x_amax = None if xa is None else xa.data_ptr()
            q.dy_amax = None if dya is None else dya.data_ptr()
            q.ldx, q.lddy, q.N, q.H, q.W, q.C, q.K, q.R, q.S, q.stride, q.pad, q.dil = ldx, lddy, N, H, W, Cc, K, R, S, stride, pad, dil
        import ctypes
        lib = _lib.load()
        like = items[0][0]
        tbytes = int(lib.dsrl_conv2d_wgrad_group_table_bytes(n))
        ws = _ws(int(lib.dsrl_conv2d_wgrad_group_workspace_bytes(ctypes.addressof(probs), n)), like)
        if graph_keepalive is not None:
            # under capture no pinned memory may be allocated (hipHostMalloc is not capturable): the table comes out of the pinned arena
            # the capturing TrainStep set aside, and that arena lives as long as the graph (its copy node re-reads it on every replay)
            global capture_host_off
            off = (capture_host_off + 255) & ~255
            if capture_host is None or off + tbytes > capture_host.numel():
                raise DsrlHipError('WgradQueue.flush under graph capture: the pinned table arena is missing or too small')
            host = capture_host[off:off + tbytes]
            capture_host_off = off + tbytes
        else:
            host = torch.empty(tbytes, dtype=torch.uint8, pin_memory=True)
        dev = torch.empty(tbytes, dtype=torch.uint8, device=like.device)
        call('dsrl_conv2d_wgrad_group_plan', ctypes.addressof(probs), n, host.data_ptr(), tbytes, dev.data_ptr(), ws.data_ptr(), ws.numel())
        dev.copy_(host, non_blocking=True)
        if graph_keepalive is not None:
            graph_keepalive.append(host)          # a captured copy re-reads this pinned buffer on every replay
        call('dsrl_conv2d_wgrad_group_launch', host.data_ptr(), dev.data_ptr(), _stream())
        for it in items:
            if it[6] is not None:
                it[6]()


wgrad_queue = None              # the open queue, if any


def open_wgrad_queue():
    global wgrad_queue
    wgrad_queue = WgradQueue() if (group_wgrad and get_conv_precision() != 'fp32') else None
    return wgrad_queue


def flush_wgrad_queue(reopen=False):
    """Launches what the open queue holds and closes it: only a step that opened a queue (FlatParams.zero_grad) defers.  `reopen`: a new
    queue takes the weight gradients of the rest of the backward pass (two-phase backward: one grouped launch set per phase)."""
    global wgrad_queue
    q, wgrad_queue = wgrad_queue, None
    if q is not None:
        q.flush()
        if reopen:
            wgrad_queue = WgradQueue()


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise DsrlHipError('dualsuperreslearningforsemseg_amd ops run on the HIP device only (got a CPU tensor); '
                               'there is no CPU fallback - move the model and inputs to cuda')


def _f32(t):
    if t.dtype != torch.float32:
        raise DsrlHipError(f'fp32 tensors expected, got {t.dtype}')
    return t


def new_cl(shape, like):
    return torch.empty(shape, device=like.device, dtype=torch.float32, memory_format=CL)


# Workspaces: kernels of one stream run one after the other and each is done with its scratch when the next starts, so ONE
# growing buffer per (device, stream) serves every call on that stream (saves ~700 allocator round trips per step).
_ws_pool = {}


_use_ws_pool = os.environ.get('DSRL_WS_POOL', '0') != '0'       # measured: no difference; the caching allocator is already cheap


_ws_none = {}


def _ws(nbytes, like):
    nbytes = int(nbytes)
    if nbytes <= 0:             # nothing to hand over: one persistent 256-byte placeholder per device instead of an allocator round trip
        buf = _ws_none.get(like.device)
        if buf is None:
            buf = _ws_none[like.device] = torch.empty(256, device=like.device, dtype=torch.uint8)
        return buf
    nbytes = max(nbytes, 256)
    if not _use_ws_pool:
        return torch.empty(nbytes, device=like.device, dtype=torch.uint8)
    key = (like.device, torch.cuda.current_stream(like.device).cuda_stream)
    buf = _ws_pool.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = _ws_pool[key] = torch.empty(max(nbytes, 64 << 20) * (2 if buf is not None else 1), device=like.device, dtype=torch.uint8)
    return buf


_query_cache = {}


def cquery(name, *args):
    """workspace-size style queries are pure functions of the shape: memoise them"""
    key = (name, args)
    v = _query_cache.get(key)
    if v is None:
        v = _query_cache[key] = query(name, *args)
    return v


def _ld_of(t):
    """Pixel stride if `t` (N,C,H,W) is pixel-major (channel stride 1, dense pixels with stride ld), else None."""
    N, C, H, W = t.shape
    st = t.stride()
    if C > 1 and st[1] != 1:
        return None
    if W > 1:
        ld = st[3]
    elif H > 1:
        ld = st[2]
    elif N > 1:
        ld = st[0]
    else:
        ld = C
    if ld < C:
        return None
    if (W > 1 and st[3] != ld) or (H > 1 and st[2] != W * ld) or (N > 1 and st[0] != H * W * ld):
        return None
    return ld


def pm(t):
    """-> (tensor, ld): a pixel-major view of `t` (copying into channels_last only when needed)."""
    _need_gpu(t); _f32(t)
    if t.dim() != 4:
        raise DsrlHipError(f'4-D (N,C,H,W) tensor expected, got shape {tuple(t.shape)}')
    ld = _ld_of(t)
    if ld is None or (t.data_ptr() % 4) != 0:
        out = new_cl(tuple(t.shape), t)
        out.copy_(t)
        return out, t.shape[1]
    return t, ld


def pm_dense(t):
    """pixel-major with ld == C"""
    t, ld = pm(t)
    if ld != t.shape[1]:
        out = new_cl(tuple(t.shape), t)
        N, Cc, H, W = t.shape
        call('dsrl_copy2d', t.data_ptr(), ld, out.data_ptr(), Cc, N * H * W, Cc, _stream())
        return out
    return t


def pm_vec4(t):
    """pixel-major view the float4 loaders of the MFMA kernels accept: 16-byte aligned base, ld a multiple of 4;
    a channel count that is not a multiple of 4 (the 19 class logits) is copied into a zero-padded buffer."""
    t, ld = pm(t)
    N, Cc, H, W = t.shape
    if Cc % 4 == 0 and ld % 4 == 0 and t.data_ptr() % 16 == 0:
        return t, ld
    cp = (Cc + 3) & ~3
    buf = new_cl((N, cp, H, W), t)
    if cp != Cc:
        buf.zero_()
    call('dsrl_copy2d', t.data_ptr(), ld, buf.data_ptr(), cp, N * H * W, Cc, _stream())
    return buf[:, :Cc], cp


def _sink(param, like_shape=None):
    """If `param` is bound to a ddp.FlatParams arena (and has not received a gradient yet in this backward), returns the arena
    view its gradient kernel may write directly - this skips the temporary + accumulate launch autograd would otherwise need."""
    owner = getattr(param, '_dsrl_arena', None)
    if owner is None or not owner.claim(param):
        return None
    return param.grad


def _sink_flat(param, n):
    """The arena slot of `param` as the n-float output of its gradient kernel, when the parameter's memory is dense in logical order (biases,
    ConvTranspose weights, the 1xCx1x1 feature-transformer filters) and this is its first gradient of the pass - else None.  The caller
    launches the kernel into it and then calls _sunk(param): no temporary, no copy launch (they were 4-5 us each, ~10 per step)."""
    owner = getattr(param, '_dsrl_arena', None)
    if owner is None or not isinstance(param, torch.nn.Parameter) or param.grad is None:
        return None
    g = param.grad
    if not g.is_cuda or g.numel() != n or not g.is_contiguous() or not owner.claim(param):
        return None
    return g


def _sunk(param):
    param._dsrl_arena.written(param)


def _deliver(param, grad):
    """Hands a freshly computed parameter gradient to the arena instead of to autograd: when `param` lives in a ddp.FlatParams arena and this
    is its first gradient of the pass, the values are copied into its (zeroed) arena slot on the CURRENT stream, the reducer is told, and
    autograd gets nothing.  Besides saving autograd's own accumulation launch this keeps leaf accumulation off other streams: torch runs an
    AccumulateGrad node on the stream its parameter was created on, which under graph capture forks the capture across streams (the warning
    'AccumulateGrad node's stream does not match ...') - with every parameter gradient delivered here the captured step is one linear stream."""
    if grad is None or not isinstance(param, torch.nn.Parameter):
        return grad
    owner = getattr(param, '_dsrl_arena', None)
    if owner is None or not grad.is_cuda or tuple(grad.shape) != tuple(param.shape) or not owner.claim(param):
        return grad
    param.grad.copy_(grad)
    owner.written(param)
    return None


def _weight_amax(w):
    """Magnitude slot of a conv filter that ddp.FlatParams keeps (measured by the batched filter transpose of this step), or None."""
    arena, slot = getattr(w, '_dsrl_arena', None), getattr(w, '_dsrl_wamax', None)
    if slot is not None and arena is not None and arena.wt_valid:
        return slot
    return None


def _weight_split(w, attr):
    """The pre-split form of a conv filter that ddp.FlatParams wrote in this step (attr: '_dsrl_wsplit' forward layout, '_dsrl_wtsplit' transposed), or None."""
    arena = getattr(w, '_dsrl_arena', None)
    return getattr(w, attr, None) if (arena is not None and arena.split_valid) else None


def _weight_planes(w, attr):
    """The fp16 planes of a conv filter that ddp.FlatParams wrote in this step (attr: '_dsrl_wplanes' [K][R][S][C], '_dsrl_wtplanes' [C][R][S][K]), or None."""
    arena = getattr(w, '_dsrl_arena', None)
    return getattr(w, attr, None) if (arena is not None and getattr(arena, 'planes_valid', False)) else None


def split_filter(w):
    """(amax record, w_split, wt_split) of one [K][R][S][C] filter for the f16x3 kernels: what ddp.FlatParams prepares for every filter of a
    model once per step (dsrl_conv2d_transpose_filters_batched + dsrl_conv2d_split_filters_batched), here for a single tensor."""
    w = w_cl(w)
    K, C, R, S = w.shape
    Kp, RS, ct = (K + 3) & ~3, R * S, (C + 31) // 32
    tiles = RS * ct * ((Kp + 31) // 32)
    rec = amax_slot(w.device)
    wt = torch.empty(C * RS * Kp, device=w.device, dtype=torch.float32)
    wsp, wtsp = torch.empty(K * RS * C, device=w.device, dtype=torch.float32), torch.empty_like(wt)
    t1 = torch.tensor([[w.data_ptr(), wt.data_ptr(), K, Kp, RS, C, 0, ct, rec.data_ptr(), 0]], dtype=torch.int64, device=w.device)
    t2 = torch.tensor([[w.data_ptr(), wtsp.data_ptr(), K, Kp, RS, C, 0, ct, rec.data_ptr(), wsp.data_ptr()]], dtype=torch.int64, device=w.device)
    call('dsrl_conv2d_transpose_filters_batched', t1.data_ptr(), 1, tiles, _stream())
    call('dsrl_conv2d_split_filters_batched', t2.data_ptr(), 1, tiles, _stream())
    return rec, wsp, wtsp, wt


def planes_of(data, ld, amax, nplanes=2):
    """fp16 planes (include/dsrl_hip.h: dsrl_split_planes) of the pixel-major tensor `data` (N,C,H,W; pixel stride ld), scaled by the amax record
    `amax`: a uint8 buffer holding [P][ld] fp16 first terms and, dsrl_planes_lo_offset(P * ld) bytes further, the second terms."""
    N, Cc, H, W = data.shape
    P = N * H * W
    buf = torch.empty(int(cquery('dsrl_planes_bytes', P * ld, nplanes)), device=data.device, dtype=torch.uint8)
    call('dsrl_split_planes', data.data_ptr(), ld, P, Cc, amax.data_ptr(), buf.data_ptr(), nplanes, _stream())
    return buf


# fp16 planes of activation operands (round 4): 'off' = never, 'all' = a standalone split pass for every eligible operand that does not carry
# planes yet, 'auto' (default) = the split pass only where it pays by itself (tensors of >= DSRL_PLANES_MIN_ELEMS elements: the 65536-pixel decoder
# operands; a split launch costs ~6 us on the small ones, more than the planes kernel gains there) - producers that write planes beside their
# fp32 output are used in every mode but 'off'
planes_mode = os.environ.get('DSRL_PLANES_MODE', 'auto')
if planes_mode == 'off':
    os.environ.setdefault('DSRL_PLANES', '0')       # the library's planner then keeps the f16x3 forward off the 256x256 tile, whose register-staged build spills
planes_min_elems = int(os.environ.get('DSRL_PLANES_MIN_ELEMS', str(8 << 20)))
# Round 5: the reduced-precision 'f16x1' arithmetic (apex O1 / O2) takes ONE plane per operand - the conv a 2-byte storage format would run.  A one-plane split
# pass moves 6 instead of 8 bytes per element and the conv gains more (tools/fp16_slice.py at config 5's shapes: cat_conv.0 forward 593 -> 406 us, dgrad
# 772 -> 547 us; layer3 3x3 48 -> 35 us), so the pass pays from 2 Mi elements on, and for the data gradients of the >= 8 Mi-element operands as well.
planes_min_elems_x1 = int(os.environ.get('DSRL_PLANES_MIN_ELEMS_X1', str(2 << 20)))


def _planes_npl():
    """planes per operand of the current arithmetic: 2 (f16x3), 1 (f16x1), 0 (no plane operands)"""
    return {4: 2, 5: 1}.get(_conv_precision_code(), 0)


def planes_wanted(data, ld, C, K, taps=9):
    """Would a conv with this activation operand take planes if its filter had them? (planes_mode 'auto' / 'all'; both channel counts multiples of 8;
    'auto': long K loops over large operands only - the 3x3 decoder convs)"""
    npl = _planes_npl()
    if planes_mode == 'off' or not npl or C % 8 or K % 8 or ld % 8 or data.data_ptr() % 16:
        return False
    N, Cc, H, W = data.shape
    return planes_mode == 'all' or (N * H * W * Cc >= (planes_min_elems if npl == 2 else planes_min_elems_x1) and taps >= 9)


def planes_for(t, data, ld, amax, taps=9, split_ok=True):
    """fp16 planes of operand tensor `t` (pixel-major copy `data`, pixel stride ld) scaled by ITS record `amax`, or None: the planes the tensor
    carries (left by its producer or by an earlier consumer of the same tensor in this step), else a split pass when planes_mode allows one."""
    npl = _planes_npl()
    if planes_mode == 'off' or amax is None or not npl:
        return None
    gen = getattr(amax, '_dsrl_gen', -1)          # a step-arena record is re-issued (same address) by the next step: its generation is part of the key
    have = getattr(t, '_dsrl_planes', None)
    if (have is not None and have[1] == ld and have[2] == amax.data_ptr() and have[3] == data.data_ptr() and have[4] == t._version and have[5] == gen and
            have[6] >= npl):
        return have[0]
    N, Cc, H, W = data.shape
    if Cc % 8 or ld % 8 or data.data_ptr() % 16:
        return None
    if planes_mode != 'all' and (N * H * W * Cc < (planes_min_elems if npl == 2 else planes_min_elems_x1) or taps < 9 or not split_ok):
        return None
    buf = planes_of(data, ld, amax, npl)
    try:
        t._dsrl_planes = (buf, ld, amax.data_ptr(), data.data_ptr(), t._version, gen, npl)
        _planes_holders.append(weakref.ref(t))
    except Exception:           # noqa: BLE001
        pass
    return buf


# Tensors that carry cached planes.  The planes are as large as the fp32 tensor, the tensor is saved for backward, and in 'auto' mode no backward
# kernel reads x planes (weight gradients read fp32 x): without a release every large decoder operand would keep a dead full-size copy until its
# backward node ran (~1.3 GB for the 304-channel concat at 512x1024, B=8: ADVICE round 4).  drop_planes() runs when the forward consumers are
# done - at the root of backward (fused_losses_backward) and at the end of the step.
_planes_holders = []


def drop_planes():
    for r in _planes_holders:
        t = r()
        if t is not None and hasattr(t, '_dsrl_planes'):
            try:
                del t._dsrl_planes
            except Exception:           # noqa: BLE001
                pass
    _planes_holders.clear()


def filter_planes(w, rec):
    """(w_planes [K][R][S][C], wt_planes [C][R][S][K]) of one filter, scaled by its amax record `rec`: what ddp.FlatParams prepares for every
    filter once per step (dsrl_conv2d_filter_planes_batched), here for a single tensor.  K and C must be multiples of 8."""
    w = w_cl(w)
    K, C, R, S = w.shape
    RS, ct = R * S, (C + 31) // 32
    tiles = RS * ct * ((K + 31) // 32)
    nbytes = int(cquery('dsrl_planes_bytes', K * RS * C, 2))
    wp = torch.empty(nbytes, device=w.device, dtype=torch.uint8)
    wtp = torch.empty(nbytes, device=w.device, dtype=torch.uint8)
    t = torch.tensor([[w.data_ptr(), wtp.data_ptr(), K, K, RS, C, 0, ct, rec.data_ptr(), wp.data_ptr()]], dtype=torch.int64, device=w.device)
    call('dsrl_conv2d_filter_planes_batched', t.data_ptr(), 1, tiles, _stream())
    return wp, wtp


def _is_krsc(w):
    """Is the (K,C,R,S) filter physically [K][R][S][C]?  (torch does not call plain-strided 1x1 filters channels_last although the
    two layouts coincide for them)"""
    return w.is_contiguous(memory_format=CL) or (w.shape[2] == 1 and w.shape[3] == 1 and w.is_contiguous())


def w_cl(w):
    _need_gpu(w); _f32(w)
    return w if _is_krsc(w) else w.contiguous(memory_format=CL)


def _out_size(n, k, stride, pad, dil):
    return (n + 2 * pad - dil * (k - 1) - 1) // stride + 1


_rng_state = {'seed': 0x5EED, 'step': 0, 'current': 0}


def set_dropout_seed(seed):
    """Base key of the Philox dropout generator; every forward pass (begin_forward) derives its own key from it."""
    _rng_state['seed'] = int(seed) & 0xFFFFFFFFFFFF
    _rng_state['step'] = 0


def _derive(step):
    return (_rng_state['seed'] * 1000003 + step) & 0xFFFFFFFFFFFFFFFF


def peek_next_seed():
    return _derive(_rng_state['step'] + 1)


def begin_forward(training=True):
    """Called once per model forward: all Dropout modules of that pass share one key and differ by stream id.  An evaluation pass draws no mask
    and does not advance the key (torch's generator is not consumed by eval-mode Dropout either): the key sequence of the training steps is the
    same with and without validation passes in between, and the same for eager launches and for a graph whose key lives on the device."""
    if not training:
        return _rng_state['current']
    _rng_state['step'] += 1
    _rng_state['current'] = _derive(_rng_state['step'])
    return _rng_state['current']


def current_seed():
    return _rng_state['current']


# ------------------------------------------------------------------------------------------------ shared gradient buffers
# A tensor that feeds two branches (the input of a ResNet bottleneck: conv1 and the residual add / the downsample conv) receives two
# gradient contributions that autograd would sum with an extra elementwise kernel.  With a GradSlot the first contribution's buffer
# is published and the later data-gradient kernel accumulates into it in its epilogue (dsrl_conv2d_dgrad_accumulate) and reports
# no gradient of its own.  Only sound when EVERY consumer of the tensor takes part, hence fork(): an alias node whose output is
# consumed inside the block only; other users of the original tensor see one ordinary gradient.  Any case the protocol does not
# cover falls back to returning a separate gradient (closed slot), which autograd sums as usual.
grad_slots_enabled = os.environ.get('DSRL_GRAD_SLOTS', '1') != '0'
# one slot for the three consumers of layer1's output (ResNet101.forward / DSRL.forward_head, round 5)
outer_grad_slot = os.environ.get('DSRL_OUTER_SLOT', '1') != '0'


class GradSlot:
    __slots__ = ('buf', 'closed', 'link')

    def __init__(self):
        self.buf, self.closed = None, False
        self.link = None            # the BNLink whose backward sums the LAST contributor leaves (set where both are created); a contributor without
                                    # that link arriving after the sums exist withdraws them (they would miss its part of the gradient)


# Backward counterpart of conv2d_bn_act: when y = relu(bn(x)) feeds exactly one conv, that conv's data-gradient kernel produces the
# gradient of y in registers and can leave the BatchNorm-backward partial sums with it (dsrl_conv2d_dgrad_bnstats); the BN backward
# then runs as one streaming kernel (dsrl_bn_bwd_from_stats).  A BNLink carries what the conv needs from the BN's forward and the
# partials back; the BN only trusts them if the gradient it receives is the very buffer that conv wrote.
bn_bwd_stats_enabled = os.environ.get('DSRL_BN_BWD_STATS', '1') != '0'
# the same for bn3 via the next block's accumulating dgrad.  Rounds 3-4: correct but slower (the wide 1x1 data gradients spent 11-17 us in an epilogue of
# 4-byte loads of x / y).  Round 5: on since that epilogue reads float4 rows through LDS (conv_split_kernel.h, bn_fast): +0.65 % step throughput, 29 of the 45
# device-wide-barrier BatchNorm launches become streaming from-sums launches (profiles/round5_ab.txt)
bn_bwd_stats_shared = os.environ.get('DSRL_BN_BWD_STATS_SHARED', '1') != '0'


class BNLink:
    __slots__ = ('x', 'y_ptr', 'y_shape', 'mean', 'invstd', 'relu', 'valid', 'stats', 'parts', 'dx_ptr', 'shared', 'drop_p')

    def __init__(self, shared=False):
        self.x = self.mean = self.invstd = self.stats = None      # the BN output itself is NOT kept (it may carry this link: no cycles)
        self.y_ptr, self.y_shape = 0, None
        self.relu, self.valid, self.parts, self.dx_ptr = False, False, 0, 0
        self.shared = shared        # y feeds the two consumers of a GradSlot: the accumulating (= last) dgrad completes its gradient
        self.drop_p = 0.0           # a Dropout(p) behind the BN's ReLU (round 5): y is the tensor behind it, its zeros are the combined mask


class _Fork(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g


def fork(x):
    y = _Fork.apply(x)
    a = carried_amax(x)
    if a is not None:
        set_amax(y, a)              # the alias holds the same values: it keeps the magnitude record of the tensor it aliases
    return y


# ------------------------------------------------------------------------------------------------ conv2d
class _Conv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, dil, gslot=None, stats_parts=0, in_link=None):
        ctx.set_materialize_grads(False)       # no zero tensor for the (non-differentiable) statistics output's gradient
        ctx.gslot = gslot
        ctx.in_link = in_link
        x_in = x
        x, ldx = pm_vec4(x)
        w_param = w
        w = w_cl(w)
        N, Cc, H, W = x.shape
        K, Cw, R, S = w.shape
        if Cw != Cc:
            raise DsrlHipError(f'conv2d: input has {Cc} channels, weight expects {Cw}')
        Ho, Wo = _out_size(H, R, stride, pad, dil), _out_size(W, S, stride, pad, dil)
        y = new_cl((N, K, Ho, Wo), x)
        shp = (N, H, W, Cc, K, R, S, stride, pad, dil)
        ws = _ws(cquery('dsrl_conv2d_fwd_workspace_bytes', *shp), x)
        if bias is not None:
            _need_gpu(bias)
        stats = None
        xa = wa = None
        if f16_mode():              # operand magnitudes of the f16x3 arithmetic (the library measures what it is not given)
            xa = amax_for(x_in, x, ldx)
            wa = _weight_amax(w_param)
        ctx.amax = (xa, wa)
        if stats_parts > 0:         # BatchNorm partials of y from the conv epilogue (include/dsrl_hip.h: dsrl_conv2d_fwd_stats)
            stats = torch.empty(cquery('dsrl_bn_stats_floats', 3, int(stats_parts), K), device=x.device, dtype=torch.float32)
        wsp = _weight_split(w_param, '_dsrl_wsplit') if wa is not None else None        # the filter pre-split by ddp.FlatParams (same step, same record)
        wpl = _weight_planes(w_param, '_dsrl_wplanes') if wa is not None else None       # ... and as fp16 planes; with planes of x the launch stages by LDS-DMA
        if wpl is None and wa is not None and planes_wanted(x, ldx, Cc, K, R * S):
            w_param._dsrl_want_planes = True            # ddp.FlatParams writes this filter's planes from the next step on
        xp = planes_for(x_in, x, ldx, xa, R * S) if wpl is not None else None
        call('dsrl_conv2d_fwd_planes', x.data_ptr(), ldx, None if xa is None else xa.data_ptr(), None if xp is None else xp.data_ptr(), w.data_ptr(),
             None if wa is None else wa.data_ptr(), None if wsp is None else wsp.data_ptr(), None if wpl is None else wpl.data_ptr(),
             None if bias is None else bias.data_ptr(), y.data_ptr(), K, *shp, ws.data_ptr(), ws.numel(),
             None if stats is None else stats.data_ptr(), int(stats_parts), _stream())
        ctx.save_for_backward(x, w)
        ctx.shp = shp
        ctx.has_bias = bias is not None
        ctx.bparam = bias if isinstance(bias, torch.nn.Parameter) else None
        ctx.wparam = w_param if isinstance(w_param, torch.nn.Parameter) else None
        ctx.wt = None
        if pretranspose_filters and ctx.needs_input_grad[0]:
            # the data-gradient kernel reads the filter transposed: build that copy now, on the side stream, off the critical path
            cur, side = torch.cuda.current_stream(), side_stream(x.device)
            side.wait_stream(cur)                                   # the weights (last SGD update) are ready on the compute stream
            with torch.cuda.stream(side):
                wt = torch.empty(cquery('dsrl_conv2d_transposed_filter_floats', Cc, K, R, S), device=x.device, dtype=torch.float32)
                call('dsrl_conv2d_transpose_filter', w.data_ptr(), wt.data_ptr(), Cc, K, R, S, side.cuda_stream)
                ev = torch.cuda.Event()
                ev.record(side)
            w.record_stream(side)
            ctx.wt = (wt, ev)
        if stats is not None:
            ctx.mark_non_differentiable(stats)
            return y, stats
        return y

    @staticmethod
    def backward(ctx, dy, *_unused):
        if dy is None:
            return (None,) * 9
        x, w = ctx.saved_tensors
        shp = ctx.shp
        N, H, W, Cc, K, R, S, stride, pad, dil = shp
        dy_in = dy
        dy, lddy = pm_vec4(dy)
        ldx = _ld_of(x)
        dx = dw = db = None
        st = _stream()
        xa, wa = ctx.amax
        dya = None
        if f16_mode():
            dya = amax_for(dy_in, dy, lddy)          # one measurement for the data and the weight gradient
            if xa is None:
                xa = amax_for(x, x, ldx)
            if wa is not None and ctx.wparam is not None and not getattr(getattr(ctx.wparam, '_dsrl_arena', None), 'wt_valid', False):
                wa = None                             # the filter changed since its magnitude was taken
        p_ = lambda t_: None if t_ is None else t_.data_ptr()       # noqa: E731
        wtsp = _weight_split(ctx.wparam, '_dsrl_wtsplit') if (wa is not None and ctx.wparam is not None) else None
        wtpl = _weight_planes(ctx.wparam, '_dsrl_wtplanes') if (wa is not None and ctx.wparam is not None) else None
        # 'auto' never pays a split pass for a gradient (measured: the data-gradient launches gain less than the pass costs); planes a producer left are used
        # f16x1 (round 5): the one-plane split of a large gradient does pay (cat_conv.0 dgrad 772 -> 547 us at config 5's size for a ~60 us pass)
        dy_split_ok = planes_mode == 'all' or (_planes_npl() == 1 and dy.shape[0] * dy.shape[1] * dy.shape[2] * dy.shape[3] >= planes_min_elems)
        dyp = planes_for(dy_in, dy, lddy, dya, R * S, split_ok=dy_split_ok) if (wtpl is not None and ctx.needs_input_grad[0] and stride == 1 and K % 8 == 0) else None
        if ctx.needs_input_grad[1]:
            sink = _sink(ctx.wparam) if ctx.wparam is not None and _is_krsc(ctx.wparam) else None
            if sink is not None and wgrad_queue is not None:
                # deferred: all weight gradients of this backward pass run as a few grouped grids when the pass is over
                wp = ctx.wparam
                wgrad_queue.add(x, ldx, dy, lddy, sink, shp, lambda wp=wp: wp._dsrl_arena.written(wp), xa, dya)
            elif sink is not None and overlap_wgrad and ctx.needs_input_grad[0]:
                cur, side = torch.cuda.current_stream(), side_stream(x.device)
                side.wait_stream(cur)                                   # dy (and x) are ready on the compute stream
                with torch.cuda.stream(side):
                    ws = _ws(cquery('dsrl_conv2d_wgrad_workspace_bytes', *shp), x)
                    call('dsrl_conv2d_wgrad_amax', x.data_ptr(), ldx, p_(xa), dy.data_ptr(), lddy, p_(dya), sink.data_ptr(), *shp, ws.data_ptr(), ws.numel(), side.cuda_stream)
                x.record_stream(side); dy.record_stream(side)
                ctx.wparam._dsrl_arena.written(ctx.wparam, side)
            else:
                dw = sink if sink is not None else torch.empty((K, Cc, R, S), device=x.device, dtype=torch.float32, memory_format=CL)
                ws = _ws(cquery('dsrl_conv2d_wgrad_workspace_bytes', *shp), x)
                call('dsrl_conv2d_wgrad_amax', x.data_ptr(), ldx, p_(xa), dy.data_ptr(), lddy, p_(dya), dw.data_ptr(), *shp, ws.data_ptr(), ws.numel(), st)
                if sink is not None:
                    ctx.wparam._dsrl_arena.written(ctx.wparam)
                    dw = None
        if ctx.needs_input_grad[0]:
            slot = ctx.gslot
            acc = (slot is not None and not slot.closed and slot.buf is not None and tuple(slot.buf.shape) == (N, Cc, H, W)
                   and slot.buf.is_contiguous(memory_format=CL))
            dx = slot.buf if acc else new_cl((N, Cc, H, W), x)
            if slot is not None and slot.link is not None and slot.link is not ctx.in_link and slot.link.stats is not None:
                slot.link.stats = None      # sums left by an earlier contributor do not contain this gradient: that BatchNorm takes its own path
            ws = _ws(cquery('dsrl_conv2d_dgrad_workspace_bytes', *shp), x)
            wt_ptr = None
            if ctx.wt is not None:
                wt, ev = ctx.wt
                torch.cuda.current_stream().wait_event(ev)
                wt.record_stream(torch.cuda.current_stream())
                wt_ptr = wt.data_ptr()
            elif ctx.wparam is not None:
                # ddp.FlatParams keeps a transposed copy of every filter, refreshed by one batched launch per training step
                arena, wt = getattr(ctx.wparam, '_dsrl_arena', None), getattr(ctx.wparam, '_dsrl_wt', None)
                if wt is not None and arena is not None and arena.wt_valid and arena.wt_fp32_valid:
                    wt_ptr = wt.data_ptr()
            link = ctx.in_link
            parts = 0
            # only the launch that completes the gradient may produce the sums: a plain dgrad when x has this single consumer, or the
            # accumulating dgrad that adds the last contribution to a shared buffer (GradSlot with two consumers)
            final = (not acc and ctx.gslot is None) or (acc and getattr(link, 'shared', False))
            if (link is not None and link.valid and final and link.y_ptr == x.data_ptr() and link.y_shape == (N, Cc, H, W) and Cc % 32 == 0):
                parts = int(query('dsrl_conv2d_dgrad_stats_parts', *shp))
            if parts > 0:
                # x is y = relu(bn(.)) of a BatchNorm that feeds only this conv: leave its backward partial sums with the data gradient
                bstats = torch.empty(cquery('dsrl_bn_stats_floats', 2, parts, Cc), device=x.device, dtype=torch.float32)
                _, bld = pm(link.x)
                call('dsrl_conv2d_dgrad_planes_drop', dy.data_ptr(), lddy, p_(dya), p_(dyp), w.data_ptr(), wt_ptr, p_(wa), p_(wtsp), p_(wtpl), dx.data_ptr(), Cc, *shp,
                     ws.data_ptr(), ws.numel(), link.x.data_ptr(), bld, x.data_ptr(), ldx, link.mean.data_ptr(), link.invstd.data_ptr(), int(link.relu),
                     float(link.drop_p), bstats.data_ptr(), parts, int(acc), st)
                link.stats, link.parts, link.dx_ptr = bstats, parts, dx.data_ptr()
            else:
                call('dsrl_conv2d_dgrad_planes', dy.data_ptr(), lddy, p_(dya), p_(dyp), w.data_ptr(), wt_ptr, p_(wa), p_(wtsp), p_(wtpl), dx.data_ptr(), Cc, *shp,
                     ws.data_ptr(), ws.numel(), None, 0, None, 0, None, None, 0, None, 0, int(acc), st)
            if acc:
                dx = None                   # the contribution went into the buffer autograd already holds for this input
            elif slot is not None:
                if slot.buf is None:
                    slot.buf = dx
                else:
                    slot.closed = True      # a separate gradient is on its way to autograd: nobody may touch the published buffer any more
        if ctx.has_bias and ctx.needs_input_grad[2]:
            P = dy.shape[0] * dy.shape[2] * dy.shape[3]
            bsink = _sink_flat(ctx.bparam, K)
            db = bsink if bsink is not None else torch.empty(K, device=x.device, dtype=torch.float32)
            ws = _ws(cquery('dsrl_colsum_workspace_bytes', P, K), x)
            call('dsrl_colsum', dy.data_ptr(), lddy, P, K, db.data_ptr(), ws.data_ptr(), ws.numel(), st)
            if bsink is not None:
                _sunk(ctx.bparam); db = None
            else:
                db = _deliver(ctx.bparam, db)
        if dw is not None:
            dw = _deliver(ctx.wparam, dw)
        return dx, dw, db, None, None, None, None, None, None


class _StemConv(torch.autograd.Function):
    """kernel-R x kernel-S, stride-s conv of a few-channel image (the 7x7/2 RGB stem, ResNet101.py:28) as a row-folded
    implicit GEMM: the image is zero-padded physically to 4 channels and `pad` pixels, and the S (rounded up to 8) horizontal
    taps x 4 channels of one filter row form one contiguous 32-float K chunk."""

    @staticmethod
    def forward(ctx, x, w, stride, pad):
        _need_gpu(x, w); _f32(x); _f32(w)
        N, Cc, H, W = x.shape
        K, _, R, S = w.shape
        Sp = (S + 7) & ~7 if S > 4 else 4
        Cf = Sp * 4
        Ho, Wo = _out_size(H, R, stride, pad, 1), _out_size(W, S, stride, pad, 1)
        Hp = max(H + 2 * pad, (Ho - 1) * stride + R)
        Wp = max(W + 2 * pad, (Wo - 1) * stride + Sp) + 1
        st = _stream()
        xp = torch.empty((N, Hp, Wp, 4), device=x.device, dtype=torch.float32)
        sn, sc, sh, sw = x.stride()
        call('dsrl_pad_image_nhwc', x.data_ptr(), sn, sc, sh, sw, xp.data_ptr(), N, Cc, H, W, 4, pad, pad, Hp, Wp, st)
        # zero-padded [K][R][Sp][4] copy of the filter: the padding never changes, so the buffer is kept with the weight and only the filter taps are
        # re-copied (one launch instead of a fill + a copy per step)
        w2 = getattr(w, '_dsrl_stem_pad', None)
        if w2 is None or tuple(w2.shape) != (K, R, Sp, 4) or w2.device != x.device:
            w2 = torch.zeros((K, R, Sp, 4), device=x.device, dtype=torch.float32)
            if isinstance(w, torch.nn.Parameter):
                w._dsrl_stem_pad = w2
        w2[:, :, :S, :Cc] = w.detach().permute(0, 2, 3, 1)
        y = new_cl((N, K, Ho, Wo), x)
        macs = cquery('dsrl_conv2d_inbounds_macs', N, H, W, Cc, K, R, S, stride, pad, 1)
        shp = (N, Hp, Wp, Cf, K, R, stride, Ho, Wo)
        ws = _ws(cquery('dsrl_conv2d_rowfold_fwd_workspace_bytes', *shp), x)
        call('dsrl_conv2d_rowfold_fwd', xp.data_ptr(), 4, w2.data_ptr(), None, y.data_ptr(), K, *shp, macs, ws.data_ptr(), ws.numel(), st)
        ctx.save_for_backward(xp)
        ctx.cfg = (shp, macs, tuple(w.shape), Sp)
        ctx.wparam = w if isinstance(w, torch.nn.Parameter) else None
        return y

    @staticmethod
    def backward(ctx, dy):
        xp, = ctx.saved_tensors
        shp, macs, wshape, Sp = ctx.cfg
        K, Cc, R, S = wshape
        if ctx.needs_input_grad[0]:
            raise DsrlHipError('the image stem has no input-gradient kernel (the reference never differentiates w.r.t. the image)')
        dy, lddy = pm_vec4(dy)
        dw2 = torch.empty((K, R, Sp, 4), device=dy.device, dtype=torch.float32)
        ws = _ws(cquery('dsrl_conv2d_rowfold_wgrad_workspace_bytes', *shp), dy)
        call('dsrl_conv2d_rowfold_wgrad', xp.data_ptr(), 4, dy.data_ptr(), lddy, dw2.data_ptr(), *shp, macs, ws.data_ptr(), ws.numel(), _stream())
        dw = _deliver(ctx.wparam, dw2[:, :, :S, :Cc].permute(0, 3, 1, 2))
        return None, dw, None, None


def conv2d(x, weight, bias=None, stride=1, padding=0, dilation=1, grad_slot=None, in_link=None):
    """nn.Conv2d arithmetic (square stride/padding/dilation) on the MFMA implicit-GEMM kernels."""
    if grad_slot is not None and x.shape[1] % 4 != 0:
        grad_slot.closed = True             # the padded-channel fallback reports its own gradient: the shared buffer must not be used
        grad_slot = None
    if x.shape[1] < 4 and bias is None and dilation == 1 and not x.requires_grad:
        return _StemConv.apply(x, weight, int(stride), int(padding))
    if x.shape[1] % 4 != 0:
        padc = 4 - x.shape[1] % 4           # generic fallback: pad input and filter channels to a multiple of 4 (zeros contribute nothing)
        x = torch.nn.functional.pad(x, (0, 0, 0, 0, 0, padc))
        weight = torch.nn.functional.pad(weight, (0, 0, 0, 0, 0, padc))
    if in_link is not None and x.shape[1] % 4 == 0 and bn_bwd_stats_enabled:
        return _Conv2d.apply(x, weight, bias, int(stride), int(padding), int(dilation), grad_slot, 0, in_link)
    return _Conv2d.apply(x, weight, bias, int(stride), int(padding), int(dilation), grad_slot)


# ------------------------------------------------------------------------------------------------ BatchNorm (+res, relu, dropout)
class _BNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, relu, drop_p, seed, rng_stream, residual, rslot=None,
                stats=None, stats_parts=0, out_link=None, res_link=None):
        ctx.rslot = rslot
        ctx.out_link = out_link
        # the residual is the output of a BatchNorm without ReLU (downsample branch) that feeds only this add: our backward leaves that BatchNorm's sums
        ctx.res_link = res_link if (res_link is not None and res_link.valid and not res_link.relu and residual is not None
                                    and res_link.y_ptr == residual.data_ptr() and res_link.y_shape == tuple(x.shape)) else None
        x, ldx = pm(x)
        _need_gpu(gamma, beta, running_mean, running_var)
        N, Cc, H, W = x.shape
        P = N * H * W
        st = _stream()
        y = new_cl((N, Cc, H, W), x)
        # f16x3 conv arithmetic: the kernel leaves max |y| while it writes y (functional.amax_for finds it on the tensor)
        ya = amax_slot(x.device) if f16_mode() else None
        ya_ptr = None if ya is None else ya.data_ptr()
        res_ptr, ldr = None, 0
        if residual is not None:
            residual, ldr = pm(residual)
            res_ptr = residual.data_ptr()
        if training:
            if P <= 1:
                raise ValueError(f'Expected more than 1 value per channel when training, got input size {tuple(x.shape)}')
            ws = _ws(cquery('dsrl_bn_workspace_bytes', P, Cc), x)
            mean = torch.empty(Cc, device=x.device, dtype=torch.float32)
            invstd = torch.empty_like(mean)
            if stats is not None and stats_parts > 0:
                call('dsrl_bn_train_fwd_from_stats', x.data_ptr(), ldx, y.data_ptr(), Cc, P, Cc, float(eps), float(momentum), mean.data_ptr(), invstd.data_ptr(),
                     None if running_mean is None else running_mean.data_ptr(), None if running_var is None else running_var.data_ptr(),
                     gamma.data_ptr(), beta.data_ptr(), res_ptr, ldr, int(relu), float(drop_p), int(seed), int(rng_stream),
                     stats.data_ptr(), int(stats_parts), ya_ptr, st)
            else:
                call('dsrl_bn_train_fwd', x.data_ptr(), ldx, y.data_ptr(), Cc, P, Cc, float(eps), float(momentum), mean.data_ptr(), invstd.data_ptr(),
                     None if running_mean is None else running_mean.data_ptr(), None if running_var is None else running_var.data_ptr(),
                     gamma.data_ptr(), beta.data_ptr(), res_ptr, ldr, int(relu), float(drop_p), int(seed), int(rng_stream),
                     ws.data_ptr(), ws.numel(), ya_ptr, st)
        else:
            mean = running_mean
            invstd = torch.empty(Cc, device=x.device, dtype=torch.float32)
            call('dsrl_bn_invstd_from_var', running_var.data_ptr(), Cc, float(eps), invstd.data_ptr(), st)
            call('dsrl_bn_apply', x.data_ptr(), ldx, y.data_ptr(), Cc, P, Cc, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                 res_ptr, ldr, int(relu), float(drop_p), int(seed), int(rng_stream), ya_ptr, st)
        if ya is not None:
            set_amax(y, ya)
        ctx.save_for_backward(x, y, mean, invstd, gamma)
        if out_link is not None and (drop_p == 0.0 or relu) and Cc % 32 == 0 and ldx == Cc:
            out_link.x, out_link.mean, out_link.invstd, out_link.relu, out_link.valid = x, mean, invstd, bool(relu), True
            out_link.y_ptr, out_link.y_shape, out_link.drop_p = y.data_ptr(), tuple(y.shape), float(drop_p)
        ctx.cfg = (bool(training), bool(relu), float(drop_p), residual is not None)
        ctx.gb = (gamma, beta) if isinstance(gamma, torch.nn.Parameter) and isinstance(beta, torch.nn.Parameter) else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, invstd, gamma = ctx.saved_tensors
        training, relu, drop_p, has_res = ctx.cfg
        dy, lddy = pm(dy)
        _, ldx = pm(x)
        N, Cc, H, W = x.shape
        P = N * H * W
        dx = new_cl((N, Cc, H, W), x)
        dres = new_cl((N, Cc, H, W), x) if has_res else None
        sg = sb = None
        if ctx.gb is not None and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]:
            sg = _sink(ctx.gb[0])
            sb = _sink(ctx.gb[1]) if sg is not None else None
        dgamma = sg if sg is not None else torch.empty(Cc, device=x.device, dtype=torch.float32)
        dbeta = sb if sb is not None else torch.empty(Cc, device=x.device, dtype=torch.float32)
        dxa = amax_slot(x.device) if f16_mode() else None       # max |dx|, left by the kernel: dx is the dy operand of the conv in front of this BN
        dxa_ptr = None if dxa is None else dxa.data_ptr()
        link = ctx.out_link
        if link is not None and link.stats is not None and link.dx_ptr == dy.data_ptr() and lddy == Cc and (drop_p == 0.0 or relu):
            # the gradient we received is the buffer the consuming conv's dgrad wrote, and it left our two per-channel sums with it
            rl = ctx.res_link
            rparts = int(query('dsrl_bn_bwd_from_stats_res_parts', P, Cc, int(link.parts))) if (rl is not None and dres is not None and bn_res_stats_enabled) else 0
            if rparts > 0:
                # ... and the masked gradient we write to dres is the output gradient of the downsample branch's BatchNorm: leave its sums as well
                rstats = torch.empty(cquery('dsrl_bn_stats_floats', 2, rparts, Cc), device=x.device, dtype=torch.float32)
                _, rld = pm(rl.x)
                call('dsrl_bn_bwd_from_stats_res', x.data_ptr(), ldx, y.data_ptr(), Cc, dy.data_ptr(), lddy, dx.data_ptr(), Cc, dres.data_ptr(), Cc, P, Cc,
                     mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), int(relu), float(drop_p), int(training),
                     link.stats.data_ptr(), int(link.parts), dxa_ptr, rl.x.data_ptr(), rld, rl.mean.data_ptr(), rl.invstd.data_ptr(), rstats.data_ptr(), rparts, _stream())
                rl.stats, rl.parts, rl.dx_ptr = rstats, rparts, dres.data_ptr()
            else:
                call('dsrl_bn_bwd_from_stats_drop', x.data_ptr(), ldx, y.data_ptr(), Cc, dy.data_ptr(), lddy, dx.data_ptr(), Cc,
                     None if dres is None else dres.data_ptr(), Cc, P, Cc, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
                     dgamma.data_ptr(), dbeta.data_ptr(), int(relu), float(drop_p), int(training), link.stats.data_ptr(), int(link.parts), dxa_ptr, _stream())
        else:
            ws = _ws(cquery('dsrl_bn_workspace_bytes', P, Cc), x)
            call('dsrl_bn_bwd', x.data_ptr(), ldx, y.data_ptr(), Cc, dy.data_ptr(), lddy, dx.data_ptr(), Cc,
                 None if dres is None else dres.data_ptr(), Cc, P, Cc, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
                 dgamma.data_ptr(), dbeta.data_ptr(), int(relu), drop_p, int(training), ws.data_ptr(), ws.numel(), dxa_ptr, _stream())
        if dxa is not None:
            set_amax(dx, dxa)
        if sg is not None:
            ctx.gb[0]._dsrl_arena.written(ctx.gb[0]); dgamma = None
        if sb is not None:
            ctx.gb[1]._dsrl_arena.written(ctx.gb[1]); dbeta = None
        if dres is not None and ctx.rslot is not None:
            if ctx.rslot.buf is None:
                ctx.rslot.buf = dres            # published: a later dgrad of the same input accumulates into it (GradSlot)
            else:
                ctx.rslot.closed = True
        return dx, dgamma, dbeta, None, None, None, None, None, None, None, None, None, dres, None, None, None, None, None


bn_res_stats_enabled = os.environ.get('DSRL_BN_RES_STATS', '1') != '0'       # 0: the downsample BatchNorm reduces its own backward sums (until round 5)


def batch_norm_act(x, bn, relu=False, drop_p=0.0, seed=0, rng_stream=0, residual=None, residual_grad_slot=None, stats=None, out_link=None, res_link=None):
    """BatchNorm2d `bn` (an nn.BatchNorm2d holding the parameters/buffers) + optional residual add, ReLU, Dropout."""
    training = bn.training or bn.running_mean is None
    if training and bn.running_mean is not None:
        bn._dsrl_batches = getattr(bn, '_dsrl_batches', 0) + 1      # flushed into num_batches_tracked by HipBatchNorm2d.state_dict
    momentum = 0.1 if bn.momentum is None else bn.momentum
    return _BNAct.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, momentum, bn.eps, relu,
                        drop_p if training or drop_p == 0.0 else 0.0, seed, rng_stream, residual, residual_grad_slot if residual is not None else None,
                        stats[0] if (stats is not None and training) else None, stats[1] if (stats is not None and training) else 0,
                        out_link if bn_bwd_stats_enabled else None, res_link if (bn_bwd_stats_enabled and residual is not None) else None)


# BatchNorm statistics from the conv epilogue: the conv that feeds a training-mode BN leaves (n, mean, M2) partials of its output, and
# the BN becomes one streaming kernel without a statistics pass (dsrl_conv2d_fwd_stats + dsrl_bn_train_fwd_from_stats)
conv_bn_stats_enabled = os.environ.get('DSRL_CONV_BN_STATS', '1') != '0'


def conv2d_bn_act(x, weight, bias, stride, padding, dilation, bn, relu=False, drop_p=0.0, seed=0, rng_stream=0, residual=None,
                  grad_slot=None, residual_grad_slot=None, in_link=None, out_link=None, res_link=None):
    """batch_norm_act(conv2d(x, ...), bn, ...) with the BN batch statistics taken from the conv epilogue when the launch can provide
    them (split-precision kernels, no split-K, <= 256 row blocks, out channels a multiple of 32)."""
    training = bn.training or bn.running_mean is None
    K, C = weight.shape[0], x.shape[1]
    if conv_bn_stats_enabled and training and x.is_cuda and C % 4 == 0 and K % 32 == 0:
        N, _, H, W = x.shape
        parts = int(query('dsrl_conv2d_fwd_stats_parts', N, H, W, C, K, weight.shape[2], weight.shape[3], int(stride), int(padding), int(dilation)))
        if parts > 0:
            y, stats = _Conv2d.apply(x, weight, bias, int(stride), int(padding), int(dilation), grad_slot, parts, in_link)
            return batch_norm_act(y, bn, relu=relu, drop_p=drop_p, seed=seed, rng_stream=rng_stream, residual=residual,
                                  residual_grad_slot=residual_grad_slot, stats=(stats, parts), out_link=out_link, res_link=res_link)
    y = (_Conv2d.apply(x, weight, bias, int(stride), int(padding), int(dilation), grad_slot, 0, in_link)
         if (in_link is not None and x.is_cuda and C % 4 == 0) else conv2d(x, weight, bias, stride, padding, dilation, grad_slot=grad_slot))
    return batch_norm_act(y, bn, relu=relu, drop_p=drop_p, seed=seed, rng_stream=rng_stream, residual=residual,
                          residual_grad_slot=residual_grad_slot, out_link=out_link, res_link=res_link)


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed, rng_stream):
        x, ldx = pm(x)
        N, Cc, H, W = x.shape
        y = new_cl((N, Cc, H, W), x)
        call('dsrl_dropout_fwd', x.data_ptr(), ldx, y.data_ptr(), Cc, N * H * W, Cc, float(p), int(seed), int(rng_stream), _stream())
        ctx.cfg = (float(p), int(seed), int(rng_stream))
        return y

    @staticmethod
    def backward(ctx, dy):
        p, seed, rng_stream = ctx.cfg
        dy, ld = pm(dy)
        N, Cc, H, W = dy.shape
        dx = new_cl((N, Cc, H, W), dy)
        call('dsrl_dropout_bwd', dy.data_ptr(), ld, dx.data_ptr(), Cc, N * H * W, Cc, p, seed, rng_stream, _stream())
        return dx, None, None, None


def dropout(x, p, training, seed, rng_stream):
    if not training or p == 0.0:
        return x
    return _Dropout.apply(x, p, seed, rng_stream)


# ------------------------------------------------------------------------------------------------ resize / pools / concat
class _UpsampleAC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Ho, Wo):
        x, ldx = pm(x)
        N, Cc, H, W = x.shape
        y = new_cl((N, Cc, Ho, Wo), x)
        call('dsrl_bilinear_ac_fwd', x.data_ptr(), ldx, y.data_ptr(), Cc, N, H, W, Cc, Ho, Wo, _stream())
        ctx.shp = (N, Cc, H, W, Ho, Wo)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, Cc, H, W, Ho, Wo = ctx.shp
        dy, ld = pm(dy)
        dx = new_cl((N, Cc, H, W), dy)
        ws = _ws(cquery('dsrl_bilinear_ac_bwd_workspace_bytes', N, H, W, Cc, Ho, Wo), dy)
        call('dsrl_bilinear_ac_bwd', dy.data_ptr(), ld, dx.data_ptr(), Cc, N, H, W, Cc, Ho, Wo, ws.data_ptr(), ws.numel(), _stream())
        return dx, None, None


def upsample_bilinear_ac(x, size):
    return _UpsampleAC.apply(x, int(size[0]), int(size[1]))


class _GlobalAvgPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gslot=None):
        ctx.gslot = gslot
        x, ldx = pm(x)
        N, Cc, H, W = x.shape
        y = new_cl((N, Cc, 1, 1), x)
        call('dsrl_global_avgpool_fwd', x.data_ptr(), ldx, y.data_ptr(), N, H * W, Cc, _stream())
        ctx.shp = (N, Cc, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, Cc, H, W = ctx.shp
        dy = dy.contiguous().view(N, Cc)
        dx = new_cl((N, Cc, H, W), dy)
        call('dsrl_global_avgpool_bwd', dy.data_ptr(), dx.data_ptr(), Cc, N, H * W, Cc, _stream())
        slot = ctx.gslot
        if slot is not None and not slot.closed and slot.buf is None:
            slot.buf = dx           # first contribution of a shared input (GradSlot): the data gradients of the other consumers accumulate into it
        return dx, None


def global_avg_pool(x, grad_slot=None):
    """grad_slot: functional.GradSlot shared with the other consumers of x (ASPP: the four conv branches).  This op runs first in the backward pass
    (it is the last consumer in the forward), so its dense gradient becomes the shared buffer; published later it would stay a gradient of its own."""
    return _GlobalAvgPool.apply(x, grad_slot)


class _MaxPool3x3s2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = pm_dense(x)
        N, Cc, H, W = x.shape
        y = new_cl((N, Cc, (H - 1) // 2 + 1, (W - 1) // 2 + 1), x)
        idx = torch.empty(y.numel(), device=x.device, dtype=torch.uint8)
        call('dsrl_maxpool3x3s2_fwd', x.data_ptr(), y.data_ptr(), idx.data_ptr(), N, H, W, Cc, _stream())
        ctx.save_for_backward(idx)
        ctx.shp = (N, Cc, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        idx, = ctx.saved_tensors
        N, Cc, H, W = ctx.shp
        dy = pm_dense(dy)
        dx = new_cl((N, Cc, H, W), dy)
        call('dsrl_maxpool3x3s2_bwd', idx.data_ptr(), dy.data_ptr(), dx.data_ptr(), N, H, W, Cc, _stream())
        return dx


def max_pool3x3s2(x):
    return _MaxPool3x3s2.apply(x)


cat_one_launch = os.environ.get('DSRL_CAT_ONE_LAUNCH', '1') != '0'        # 0: a strided copy per source + a magnitude pass by the consumer (until round 5)


class _Cat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *xs):
        xs = [pm(x) for x in xs]
        N, _, H, W = xs[0][0].shape
        ctot = sum(x.shape[1] for x, _ in xs)
        y = new_cl((N, ctot, H, W), xs[0][0])
        off = 0
        st = _stream()
        n = len(xs)
        srcs = (ctypes.c_void_p * n)(*[x.data_ptr() for x, _ in xs])
        lds = (ctypes.c_int32 * n)(*[ld for _, ld in xs])
        cs = (ctypes.c_int32 * n)(*[x.shape[1] for x, _ in xs])
        if cat_one_launch and n <= 8 and query('dsrl_cat_channels_supported', srcs, lds, cs, n, y.data_ptr(), ctot, N * H * W):
            # one kernel for all sources, and it leaves the magnitude the consuming convs scale their operand by (no dsrl_amax pass over the buffer)
            ya = amax_slot(y.device) if f16_mode() else None
            call('dsrl_cat_channels', srcs, lds, cs, n, y.data_ptr(), ctot, N * H * W, None if ya is None else ya.data_ptr(), st)
            if ya is not None:
                set_amax(y, ya)
        else:
            for x, ld in xs:
                call('dsrl_copy2d', x.data_ptr(), ld, y.data_ptr() + 4 * off, ctot, N * H * W, x.shape[1], st)
                off += x.shape[1]
        ctx.sizes = [x.shape[1] for x, _ in xs]
        return y

    @staticmethod
    def backward(ctx, dy):
        outs, off = [], 0
        for c in ctx.sizes:
            outs.append(dy[:, off:off + c])       # channel-slice views (pixel-major with ld = total channels)
            off += c
        return tuple(outs)


def cat_channels(xs):
    return _Cat.apply(*xs)


# ------------------------------------------------------------------------------------------------ ConvTranspose k2s2 / PixelShuffle / pointwise
convt_ce_enabled = os.environ.get('DSRL_CONVT_CE', '1') != '0'


class LogitsGrad:
    """Hand-over between the layer that produces the logits (the last ConvTranspose2d of the SSSR decoder, DSRL.py:64-69) and the loss that is the
    root of the backward pass (fused_losses).  When the loss finds this object on its logits and the shape qualifies, it computes only the CE VALUE,
    leaves `target`, `count` (device pointer holder) here and reports no gradient; the stride-8 feature transformer leaves its incoming gradient and
    weights; the ConvTranspose backward then forms d(CE)/d(logits) (+ the transformer's rank-one term) inside its own kernel
    (dsrl_convt2x2_bwd_ce): the 319 MB gradient of the 256x512 step is neither written nor read."""
    __slots__ = ('y', 'xshape', 'x_ptr', 'armed', 'target', 'ignore_index', 'count', 'ft', 'value', 'value_key')

    def __init__(self):
        self.y = None; self.xshape = None; self.x_ptr = 0; self.armed = False; self.target = None; self.ignore_index = 255; self.count = None; self.ft = None
        self.value = None           # [CE, pixel count, ...] when the producing layer already evaluated the loss (logits_target), and for which
        self.value_key = None       # (target pointer, ignore_index, NaN flag pointer)

    def usable(self, logits, target):
        """Can the loss leave the gradient to the producer?  logits must be exactly the producer's output buffer."""
        if not (convt_ce_enabled and self.y is not None and logits.data_ptr() == self.y.data_ptr() and tuple(logits.shape) == tuple(self.y.shape)):
            return False
        N, Ci, H, W = self.xshape
        return bool(query('dsrl_convt2x2_bwd_ce_supported', self.x_ptr, logits.data_ptr(), target.data_ptr(), N, H, W, Ci, logits.shape[1]))



_logits_target = None


class logits_target:
    """with logits_target(target, ignore_index, flag): ... model(x) ... - tells the layer that produces the logits (HipConvTranspose2d.logits_layer) what
    they will be compared with, so that its forward kernel evaluates nn.CrossEntropyLoss while the output tile is on chip (dsrl_convt2x2_fwd_ce) and
    fused_losses finds the value ready (LogitsGrad.value) instead of reading the logits again.  target: (N,H,W) uint8, contiguous; flag: the int32 NaN
    flag fused_losses will be given.  Outside the block, or when the shape does not qualify, nothing changes."""

    def __init__(self, target, ignore_index, flag):
        ok = (convt_ce_enabled and target is not None and target.is_cuda and target.dtype == torch.uint8 and target.is_contiguous() and target.dim() == 3
              and flag is not None and flag.dtype == torch.int32)
        self.new = (target, int(ignore_index), flag) if ok else None

    def __enter__(self):
        global _logits_target
        self.old, _logits_target = _logits_target, self.new
        return self

    def __exit__(self, *exc):
        global _logits_target
        _logits_target = self.old
        return False


class _ConvT2x2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, holder=None):
        ctx.set_materialize_grads(False)        # dy is None when the loss left its gradient to this layer (LogitsGrad)
        ctx.holder = holder
        x = pm_dense(x)
        _need_gpu(w, bias)
        w_param = w
        w = w.contiguous()
        N, Ci, H, W = x.shape
        Co = w.shape[1]
        if w.shape[0] != Ci or tuple(w.shape[2:]) != (2, 2):
            raise DsrlHipError(f'conv_transpose2d_k2s2: weight {tuple(w.shape)} does not match input channels {Ci}')
        y = new_cl((N, Co, 2 * H, 2 * W), x)
        lt = _logits_target if holder is not None else None
        if (lt is not None and tuple(lt[0].shape) == (N, 2 * H, 2 * W) and lt[0].device == x.device
                and query('dsrl_convt2x2_fwd_ce_supported', x.data_ptr(), y.data_ptr(), N, H, W, Ci, Co)):       # (pointer arguments: not memoised)
            tgt, ign, flag = lt
            holder.value = torch.empty(8, device=x.device, dtype=torch.float32)
            holder.value_key = (tgt.data_ptr(), ign, flag.data_ptr())
            ws = _ws(cquery('dsrl_convt2x2_fwd_ce_workspace_bytes', N, H, W), x)
            call('dsrl_convt2x2_fwd_ce', x.data_ptr(), w.data_ptr(), None if bias is None else bias.data_ptr(), y.data_ptr(), N, H, W, Ci, Co,
                 tgt.data_ptr(), ign, holder.value.data_ptr(), flag.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
        else:
            call('dsrl_convt2x2_fwd', x.data_ptr(), w.data_ptr(), None if bias is None else bias.data_ptr(), y.data_ptr(), N, H, W, Ci, Co, _stream())
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.params = (w_param if isinstance(w_param, torch.nn.Parameter) else None, bias if isinstance(bias, torch.nn.Parameter) else None)
        if holder is not None:
            holder.y = y; holder.xshape = (N, Ci, H, W); holder.x_ptr = x.data_ptr()
            y._dsrl_logits_grad = holder
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        N, Ci, H, W = x.shape
        Co = w.shape[1]
        h = ctx.holder
        fused = dy is None and h is not None and h.armed
        if dy is None and not fused:
            raise DsrlHipError('conv_transpose2d_k2s2: backward reached without a gradient for the output')
        if not fused:
            dy = pm_dense(dy)
        dx = new_cl((N, Ci, H, W), x)
        wsink = _sink_flat(ctx.params[0], w.numel())
        bsink = _sink_flat(ctx.params[1], Co) if ctx.has_bias else None
        dw = wsink if wsink is not None else torch.empty_like(w)
        db = (bsink if bsink is not None else torch.empty(Co, device=x.device, dtype=torch.float32)) if ctx.has_bias else None
        ws = _ws(cquery('dsrl_convt2x2_bwd_workspace_bytes', N, H, W, Ci, Co), x)
        if fused:
            ft_g, ft_w, ft_s = h.ft if h.ft is not None else (None, None, 0)
            call('dsrl_convt2x2_bwd_ce', x.data_ptr(), w.data_ptr(), h.y.data_ptr(), h.target.data_ptr(), int(h.ignore_index), h.count.data_ptr() + 4,
                 None if ft_g is None else ft_g.data_ptr(), None if ft_w is None else ft_w.data_ptr(), int(ft_s),
                 dx.data_ptr(), dw.data_ptr(), None if db is None else db.data_ptr(), N, H, W, Ci, Co, ws.data_ptr(), ws.numel(), _stream())
            h.armed = False; h.ft = None; h.target = None; h.count = None; h.y = None; h.value = None
        else:
            call('dsrl_convt2x2_bwd', x.data_ptr(), w.data_ptr(), dy.data_ptr(), dx.data_ptr(), dw.data_ptr(), None if db is None else db.data_ptr(),
                 N, H, W, Ci, Co, ws.data_ptr(), ws.numel(), _stream())
        if wsink is not None:
            _sunk(ctx.params[0]); dw = None
        if bsink is not None:
            _sunk(ctx.params[1]); db = None
        return dx, _deliver(ctx.params[0], dw), _deliver(ctx.params[1], db), None


def conv_transpose2d_k2s2(x, weight, bias=None, logits_grad=None):
    """logits_grad: a LogitsGrad when this output may be the logits of fused_losses (see there); None otherwise."""
    return _ConvT2x2.apply(x, weight, bias, logits_grad)


class _PixelShuffle(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, r):
        x = pm_dense(x)
        N, Cc, H, W = x.shape
        c = Cc // (r * r)
        y = new_cl((N, c, H * r, W * r), x)
        call('dsrl_pixel_shuffle_fwd', x.data_ptr(), y.data_ptr(), N, H, W, c, r, _stream())
        ctx.shp = (N, Cc, H, W, c, r)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, Cc, H, W, c, r = ctx.shp
        dy = pm_dense(dy)
        dx = new_cl((N, Cc, H, W), dy)
        call('dsrl_pixel_shuffle_bwd', dy.data_ptr(), dx.data_ptr(), N, H, W, c, r, _stream())
        return dx, None


def pixel_shuffle(x, r):
    return _PixelShuffle.apply(x, int(r))


class _PointwiseStrided(torch.autograd.Function):
    """1x1 conv, stride s, one output channel, no bias (DSRL.py:88-93)."""

    @staticmethod
    def forward(ctx, x, w, stride, out_slot=None):
        ctx.out_slot = out_slot
        ctx.logits_grad = getattr(x, '_dsrl_logits_grad', None)
        x = pm_dense(x)
        _need_gpu(w)
        N, Cc, H, W = x.shape
        wf = w.contiguous().view(-1)
        if wf.numel() != Cc:
            raise DsrlHipError('pointwise_strided: weight must be (1,C,1,1)')
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        y = new_cl((N, 1, Ho, Wo), x)
        call('dsrl_pointwise_strided_fwd', x.data_ptr(), wf.data_ptr(), y.data_ptr(), N, H, W, Cc, stride, _stream())
        ctx.save_for_backward(x, wf)
        ctx.stride = stride
        ctx.wshape = tuple(w.shape)
        ctx.wparam = w if isinstance(w, torch.nn.Parameter) else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wf = ctx.saved_tensors
        N, Cc, H, W = x.shape
        dy = dy.contiguous()
        # x also feeds the fused loss (fused_losses): that node ran first (it is the root of the backward pass) and published the dense
        # gradient it returned for x; our contribution lives on the stride grid only (1/64 of the pixels) and is added into that buffer
        # instead of being materialised as a mostly-zero tensor that autograd would then add with a full pass
        slot = ctx.out_slot
        lg = ctx.logits_grad
        if lg is not None and lg.armed and lg.y is not None and lg.y.data_ptr() == x.data_ptr() and lg.ft is None:
            # the loss left d(loss)/d(x) to the layer that produced x (LogitsGrad): that kernel adds g * w on the stride grid itself
            wsink = _sink_flat(ctx.wparam, Cc)
            dw = wsink if wsink is not None else torch.empty(Cc, device=x.device, dtype=torch.float32)
            ws = _ws(cquery('dsrl_pointwise_strided_bwd_workspace_bytes', N, H, W, Cc, ctx.stride), x)
            call('dsrl_pointwise_strided_bwd', x.data_ptr(), wf.data_ptr(), dy.data_ptr(), None, dw.data_ptr(), 2,
                 N, H, W, Cc, ctx.stride, ws.data_ptr(), ws.numel(), _stream())
            lg.ft = (dy, wf, ctx.stride)
            if wsink is not None:
                _sunk(ctx.wparam)
                return None, None, None, None
            return None, _deliver(ctx.wparam, dw.view(ctx.wshape)), None, None
        acc = (slot is not None and not slot.closed and slot.buf is not None and tuple(slot.buf.shape) == (N, Cc, H, W)
               and _ld_of(slot.buf) == Cc)
        dx = slot.buf if acc else new_cl((N, Cc, H, W), x)
        wsink = _sink_flat(ctx.wparam, Cc)
        dw = wsink if wsink is not None else torch.empty(Cc, device=x.device, dtype=torch.float32)
        ws = _ws(cquery('dsrl_pointwise_strided_bwd_workspace_bytes', N, H, W, Cc, ctx.stride), x)
        call('dsrl_pointwise_strided_bwd', x.data_ptr(), wf.data_ptr(), dy.data_ptr(), dx.data_ptr(), dw.data_ptr(), int(acc),
             N, H, W, Cc, ctx.stride, ws.data_ptr(), ws.numel(), _stream())
        if slot is not None and not acc:
            slot.closed = True
        if wsink is not None:
            _sunk(ctx.wparam)
            return (None if acc else dx), None, None, None
        return (None if acc else dx), _deliver(ctx.wparam, dw.view(ctx.wshape)), None, None


def pointwise_strided(x, weight, stride, out_slot=None):
    return _PointwiseStrided.apply(x, weight, int(stride), out_slot)


# ------------------------------------------------------------------------------------------------ losses
class _CrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        logits, ld = pm(logits)
        _need_gpu(target)
        if target.dtype != torch.uint8:
            target = target.to(torch.uint8)
        target = target.contiguous()
        N, Cc, H, W = logits.shape
        P = N * H * W
        if target.numel() != P:
            raise DsrlHipError(f'cross_entropy: target has {target.numel()} pixels, logits {P}')
        out = torch.empty(2, device=logits.device, dtype=torch.float32)
        ws = _ws(cquery('dsrl_ce_workspace_bytes', P), logits)
        call('dsrl_ce_fwd', logits.data_ptr(), ld, target.data_ptr(), P, Cc, int(ignore_index), out.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
        ctx.save_for_backward(logits, target, out)
        ctx.ignore_index = int(ignore_index)
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        logits, target, out = ctx.saved_tensors
        _, ld = pm(logits)
        N, Cc, H, W = logits.shape
        g = g.reshape(1).contiguous().float()
        dl = new_cl((N, Cc, H, W), logits)
        call('dsrl_ce_bwd', logits.data_ptr(), ld, target.data_ptr(), N * H * W, Cc, ctx.ignore_index, out.data_ptr(), g.data_ptr(),
             dl.data_ptr(), Cc, _stream())
        return dl, None, None


def cross_entropy(logits, target, ignore_index=255):
    """nn.CrossEntropyLoss(ignore_index) with mean reduction; target (N,H,W) uint8/long."""
    return _CrossEntropy.apply(logits, target, ignore_index)


class _MSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a = pm_dense(a)
        b = pm_dense(b)
        if a.shape != b.shape:
            raise DsrlHipError(f'mse: shapes differ {tuple(a.shape)} vs {tuple(b.shape)}')
        n = a.numel()
        out = torch.empty(1, device=a.device, dtype=torch.float32)
        ws = _ws(cquery('dsrl_mse_workspace_bytes', n), a)
        call('dsrl_mse_fwd', a.data_ptr(), b.data_ptr(), n, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
        ctx.save_for_backward(a, b)
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.reshape(1).contiguous().float()
        da = new_cl(tuple(a.shape), a)
        call('dsrl_mse_bwd', a.data_ptr(), b.data_ptr(), a.numel(), g.data_ptr(), da.data_ptr(), _stream())
        return da, None


def mse_loss(a, b):
    return _MSE.apply(a, b)


class _FusedLosses(torch.autograd.Function):
    """vals = [CE, w1 * MSE, w2 * FA, total, NaN flag] of train_or_resume.py:435-438 with the big gradients produced in the FORWARD pass:
    one pass over the logits gives the CE value and d(CE)/d(logits), one over the SISR output the MSE value and its gradient
    (dsrl_ce_fused / dsrl_mse_fused), both also check their input for NaN.  Contract: `vals[3]` is the root of the backward pass, i.e.
    its incoming gradient is 1 (TrainStep calls vals[3].backward()); the stored gradients are returned as they are."""

    @staticmethod
    def forward(ctx, sssr, sisr, ft1, ft2, target, org, ignore_index, w1, w2, stage, flag, k):
        want = bool(ctx.needs_input_grad[0])          # forward-only (validation, no_grad): no gradient buffers are written
        logits, ld = pm(sssr)
        _need_gpu(target)
        if target.dtype != torch.uint8:
            target = target.to(torch.uint8)
        target = target.contiguous()
        N, Cc, H, W = logits.shape
        P = N * H * W
        if target.numel() != P:
            raise DsrlHipError(f'fused_losses: target has {target.numel()} pixels, logits {P}')
        st = _stream()
        dev = logits.device
        scal = torch.empty(8, device=dev, dtype=torch.float32)          # [0:2] CE + pixel count, [2] MSE
        vals = torch.empty(5, device=dev, dtype=torch.float32)
        lg = getattr(sssr, '_dsrl_logits_grad', None) if want else None
        if lg is not None and not (ld == Cc and lg.usable(logits, target)):
            lg = None
        dl = new_cl((N, Cc, H, W), logits) if (want and lg is None) else None      # lg: the producer of the logits forms this gradient in its own backward
        if lg is not None and lg.value is not None and lg.value_key == (target.data_ptr(), int(ignore_index), flag.data_ptr()):
            scal = lg.value                                             # the producer's forward kernel evaluated the loss (logits_target)
        else:
            ws = _ws(cquery('dsrl_ce_fused_workspace_bytes', P), logits)
            call('dsrl_ce_fused', logits.data_ptr(), ld, target.data_ptr(), P, Cc, int(ignore_index), None if dl is None else dl.data_ptr(), Cc,
                 scal.data_ptr(), flag.data_ptr(), ws.data_ptr(), ws.numel(), st)
        da = None
        mse_ptr = fa_ptr = None
        ctx.fa = None
        if stage > 1:
            a = pm_dense(sisr); b = pm_dense(org)
            if a.shape != b.shape:
                raise DsrlHipError(f'fused_losses: SISR output {tuple(a.shape)} vs input_org {tuple(b.shape)}')
            da = new_cl(tuple(a.shape), a) if want else None
            ws2 = _ws(cquery('dsrl_mse_workspace_bytes', a.numel()), a)
            mse_ptr = scal.data_ptr() + 8
            call('dsrl_mse_fused', a.data_ptr(), b.data_ptr(), a.numel(), float(w1), None if da is None else da.data_ptr(), mse_ptr, flag.data_ptr(),
                 ws2.data_ptr(), ws2.numel(), st)
        if stage > 2:
            _need_gpu(ft1, ft2); _f32(ft1); _f32(ft2)
            if ft1.stride() != ft2.stride() or ft1.data_ptr() % 4 or ft2.data_ptr() % 4:
                ft1, ft2 = ft1.contiguous(), ft2.contiguous()
            nan_check_(flag, ft1, ft2)
            B, Cf, Hf, Wf = ft1.shape
            fa_out = torch.empty(1, device=dev, dtype=torch.float32)
            saved = torch.empty(cquery('dsrl_fa_saved_floats', B, Cf, Hf, Wf, k), device=dev, dtype=torch.float32)
            ws3 = _ws(cquery('dsrl_fa_workspace_bytes', B, Cf, Hf, Wf, k), ft1)
            sb, sc, sh, sw = ft1.stride()
            call('dsrl_fa_fwd', ft1.data_ptr(), ft2.data_ptr(), B, Cf, Hf, Wf, sb, sc, sh, sw, k, 0, fa_out.data_ptr(), saved.data_ptr(),
                 ws3.data_ptr(), ws3.numel(), st)
            fa_ptr = fa_out.data_ptr()
            ctx.fa = (ft1, ft2, saved, k, fa_out)
        call('dsrl_loss_mix', scal.data_ptr(), mse_ptr, fa_ptr, float(w1), float(w2), flag.data_ptr(), vals.data_ptr(), st)
        if lg is not None:
            lg.target = target; lg.ignore_index = int(ignore_index); lg.count = scal; lg.ft = None; lg.armed = True
        ctx.lg = lg
        ctx.grads = (dl, da)
        ctx.w2 = float(w2)
        ctx.slots = (getattr(sssr, '_dsrl_out_slot', None), getattr(sisr, '_dsrl_out_slot', None) if stage > 1 else None)
        ctx.keep = scal
        return vals

    @staticmethod
    def backward(ctx, g):
        dl, da = ctx.grads
        if dl is None and ctx.lg is None:
            raise DsrlHipError('fused_losses: backward without a gradient-enabled forward')
        # the gradients were written by the forward pass for d(total) = 1: a caller that scales the loss or differentiates another element would get
        # them unscaled.  Checked once per process (one host read), outside graph capture - the training step's own call pattern never changes.
        global _fused_losses_root_checked
        if not _fused_losses_root_checked and not torch.cuda.is_current_stream_capturing():
            gv = g.detach().float().cpu()
            if gv.numel() != 5 or float(gv[3]) != 1.0 or float(gv[:3].abs().sum()) != 0.0 or float(gv[4]) != 0.0:
                raise DsrlHipError('fused_losses: only vals[3].backward() with unit gradient is supported (the loss gradients are formed in the forward '
                                   f'pass); got an incoming gradient of {gv.tolist()} - use functional.cross_entropy / mse_loss / FALoss for a scaled loss')
            _fused_losses_root_checked = True
        d1 = d2 = None
        if ctx.fa is not None:
            ft1, ft2, saved, k, _ = ctx.fa
            B, Cf, Hf, Wf = ft1.shape
            gw = _const1(ctx.w2, ft1.device)
            d1 = torch.empty((B, Cf, Hf, Wf), device=ft1.device, dtype=torch.float32)
            d2 = torch.empty_like(d1)
            sb, sc, sh, sw = ft1.stride()
            call('dsrl_fa_bwd', ft1.data_ptr(), ft2.data_ptr(), B, Cf, Hf, Wf, sb, sc, sh, sw, k, 0, gw.data_ptr(), saved.data_ptr(),
                 d1.data_ptr(), d2.data_ptr(), None, 0, _stream())
        # publish the two dense gradients: the stride-8 feature transformers add their sparse contributions into them (GradSlot protocol)
        for slot, buf in zip(ctx.slots, (dl, da)):
            if slot is not None and buf is not None and not slot.closed and slot.buf is None:
                slot.buf = buf
        return dl, da, d1, d2, None, None, None, None, None, None, None, None


_fused_losses_root_checked = False
_const_cache = {}


def _const1(value, device):
    """One-element fp32 device constant, created once per (value, device): a torch.full per step is a 5 us launch."""
    key = (float(value), device)
    c = _const_cache.get(key)
    if c is None:
        if torch.cuda.is_current_stream_capturing():
            return torch.full((1,), float(value), device=device, dtype=torch.float32)
        c = _const_cache[key] = torch.full((1,), float(value), device=device, dtype=torch.float32)
    return c


def fused_losses_backward(vals):
    """vals[3].backward() without autograd's select / ones / zeros launches at the root of the pass: the unit gradient of element 3 is a
    cached constant."""
    drop_planes()               # every forward consumer of this step's plane operands has been enqueued
    key = ('e3', vals.device)
    e3 = _const_cache.get(key)
    if e3 is None:
        if torch.cuda.is_current_stream_capturing():
            vals[3].backward()
            return
        e3 = _const_cache[key] = torch.tensor([0.0, 0.0, 0.0, 1.0, 0.0], device=vals.device, dtype=torch.float32)
    torch.autograd.backward([vals], [e3])


def fused_losses(outs, target, input_org, ignore_index, w1, w2, stage, flag, subsample_factor=8):
    """-> 5-float device tensor [CE, w1*MSE, w2*FA, total, NaN flag]; `outs` = DSRL.forward's 4-tuple.  vals[3].backward() is the only
    supported backward (the function is the root of the pass); `flag` is the int32 NaN flag the fused kernels OR into."""
    sssr, sisr, ft1, ft2 = outs
    dummy = _const1(0.0, sssr.device)            # stands in for the outputs a lower stage does not have (a cached constant: no fill launch per step)
    return _FusedLosses.apply(sssr, sisr if stage > 1 else dummy, ft1 if stage > 2 else dummy, ft2 if stage > 2 else dummy, target,
                              input_org if stage > 1 else dummy, int(ignore_index), float(w1), float(w2), int(stage), flag, int(subsample_factor))


_RED = {'mean': 0, 'sum': 1, 'none': 2}


class _FALoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fm1, fm2, k, reduction):
        _need_gpu(fm1, fm2); _f32(fm1); _f32(fm2)
        if fm1.stride() != fm2.stride() or fm1.data_ptr() % 4 or fm2.data_ptr() % 4:
            fm1, fm2 = fm1.contiguous(), fm2.contiguous()
        B, Cc, H, W = fm1.shape
        red = _RED[reduction]
        n = (W // k) ** 2
        out = torch.empty((B, Cc, n * n) if red == 2 else (1,), device=fm1.device, dtype=torch.float32)
        saved = torch.empty(cquery('dsrl_fa_saved_floats', B, Cc, H, W, k), device=fm1.device, dtype=torch.float32)
        ws = _ws(cquery('dsrl_fa_workspace_bytes', B, Cc, H, W, k), fm1)
        sb, sc, sh, sw = fm1.stride()
        call('dsrl_fa_fwd', fm1.data_ptr(), fm2.data_ptr(), B, Cc, H, W, sb, sc, sh, sw, k, red, out.data_ptr(), saved.data_ptr(),
             ws.data_ptr(), ws.numel(), _stream())
        ctx.save_for_backward(fm1, fm2, saved)
        ctx.cfg = (k, red)
        return out if red == 2 else out[0].clone()

    @staticmethod
    def backward(ctx, g):
        fm1, fm2, saved = ctx.saved_tensors
        k, red = ctx.cfg
        if red == 2:
            raise DsrlHipError("FALoss(reduction='none') has no backward kernel; use 'mean' or 'sum'")
        B, Cc, H, W = fm1.shape
        g = g.reshape(1).contiguous().float()
        d1 = torch.empty((B, Cc, H, W), device=fm1.device, dtype=torch.float32)
        d2 = torch.empty_like(d1)
        sb, sc, sh, sw = fm1.stride()
        call('dsrl_fa_bwd', fm1.data_ptr(), fm2.data_ptr(), B, Cc, H, W, sb, sc, sh, sw, k, red, g.data_ptr(), saved.data_ptr(),
             d1.data_ptr(), d2.data_ptr(), None, 0, _stream())
        return d1, d2, None, None


def fa_loss(fm1, fm2, subsample_factor=8, reduction='mean'):
    return _FALoss.apply(fm1, fm2, int(subsample_factor), reduction)


# ------------------------------------------------------------------------------------------------ optimiser / checks
def sgd_step_(p, g, buf, lr, momentum, weight_decay, grad_scale=1.0):
    """In-place SGD(momentum, weight_decay) on flat fp32 arenas."""
    _need_gpu(p, g, buf)
    call('dsrl_sgd_step', p.data_ptr(), g.data_ptr(), buf.data_ptr(), p.numel(), float(lr), float(momentum), float(weight_decay),
         float(grad_scale), _stream())


def sgd_step_dev_(p, g, buf, hyper):
    """The same update with (lr, momentum, weight_decay, grad_scale) read from the 4-float device tensor `hyper` at run time."""
    _need_gpu(p, g, buf, hyper)
    call('dsrl_sgd_step_dev', p.data_ptr(), g.data_ptr(), buf.data_ptr(), p.numel(), hyper.data_ptr(), _stream())


class DeviceRng:
    """Device-resident twin of the host dropout key (begin_forward): three 64-bit words {key, step, base}.  While bound, every
    dropout-bearing kernel of the device reads `key` when it runs and advance() enqueues the per-step derivation, so that a step
    captured in a hipGraph draws fresh masks on every replay (include/dsrl_hip.h: dsrl_rng_bind_device_key).  The binding is one address
    per device: advance() re-binds its own state if another DeviceRng took the device in between (a captured launch keeps the address it
    was captured with)."""
    _active = {}            # device index -> the DeviceRng whose state address is currently bound

    def __init__(self, device):
        self.state = torch.zeros(3, dtype=torch.int64, device=device)
        self.sync_from_host()
        self._bind()

    def _bind(self):
        with torch.cuda.device(self.state.device):
            call('dsrl_rng_bind_device_key', self.state.data_ptr())
        DeviceRng._active[self.state.device.index] = self

    def sync_from_host(self):
        def s64(v):
            v &= 0xFFFFFFFFFFFFFFFF
            return v - (1 << 64) if v >= (1 << 63) else v
        self.state.copy_(torch.tensor([s64(_derive(_rng_state['step'])), s64(_rng_state['step']), s64(_rng_state['seed'])], dtype=torch.int64))

    def advance(self):
        if DeviceRng._active.get(self.state.device.index) is not self:
            self._bind()
        call('dsrl_rng_advance_key', self.state.data_ptr(), _stream())

    def release(self):
        if DeviceRng._active.get(self.state.device.index) is self:
            with torch.cuda.device(self.state.device):
                call('dsrl_rng_bind_device_key', None)
            del DeviceRng._active[self.state.device.index]


def nan_check_(flag, *tensors):
    """ORs 1 into the int32 device scalar `flag` if any tensor holds a NaN (one readback for all of them)."""
    st = _stream()
    for t in tensors:
        if t is None or not t.is_cuda:
            continue
        td = t if t.is_contiguous() or t.is_contiguous(memory_format=CL) else t.contiguous()
        call('dsrl_nan_check', td.data_ptr(), td.numel(), flag.data_ptr(), st)


def conv2d_inbounds_macs(N, H, W, Cc, K, R, S, stride, pad, dil):
    return cquery('dsrl_conv2d_inbounds_macs', N, H, W, Cc, K, R, S, stride, pad, dil)
