"""Stage-1/2/3 training driver on the MI355X kernels - counterpart of the reference's
command_handlers/train_or_resume.py (same `train_or_resume(...)` signature, same seeding, SGD hyper-parameters, loss mix,
per-epoch polynomial LR, per-rank batch size, mean-reduced gradients and `final.weights` / `.checkpoint` dict formats).

What differs by design:
  * gradients are reduced and the SGD update applied on flat arenas (ddp.FlatParams) over RCCL instead of torch DDP buckets;
  * the reference's 8 blocking device->host reads and 3 host-side NaN scans per iteration (train_or_resume.py:406-451) are
    folded into ONE readback of [CE, MSE, FA, Total, nan_flag] per iteration with the same abort-on-NaN behaviour;
  * apex is not used. `mixed_precision` (apex opt levels 'O0'..'O3' in the reference, train_or_resume.py:68-72, 441-444) selects the
    arithmetic of the MFMA conv kernels instead: 'O0' = fp32-equivalent f16x3 (the default), 'O1' = bf16x6 forward / bf16x3 gradients, 'O2'/'O3' =
    bf16x3 everywhere (BASELINE config 5's reduced-precision MFMA path).  Storage and accumulation stay fp32 in every mode, so there
    is no loss scaling (nothing can underflow that fp32 training would keep) and `amp_state_dict` is None;
  * the input pipeline (torchvision Cityscapes + PIL transforms) is out of scope: `dataset` may carry a 'loader_factory'
    (callable(split, batch_size, device, rank, world) -> iterable of ((input_image, input_org), (target, _))) and
    `SyntheticCityscapes` below provides device-resident batches of the Cityscapes shapes.
"""
import glob
import os
from datetime import datetime

import torch as t
import torch.distributed as dist

from .. import functional as HF
from .. import settings
from ..ddp import FlatParams
from ..models import DSRL
from ..models.losses import FALoss


def isCUDAdevice(device):
    return device.casefold() == 'gpu'


def polynomial_lr(base_lr, end_lr, epoch, max_decay_steps, power):
    """models/schedulers/PolynomialLR.py:22-34: lr used during `epoch` (0-based); epoch 0 keeps the base lr."""
    if epoch <= 0:
        return base_lr
    return (base_lr - end_lr) * ((1. - epoch / max_decay_steps) ** power) + end_lr


class SyntheticCityscapes:
    """Device-resident synthetic batches of the shapes JointScaledImage produces (models/transforms/JointScaledImage.py:27-32):
    input_org ~ N(0,1) at the output size, input_image = align-corners bilinear resize of it to the input size,
    target uint8 in [0,19) with ~10 % of the pixels set to the ignore label."""

    def __init__(self, batch_size, input_size, device, rank=0, length=8, num_classes=19, ignore=255, distinct=1):
        g = t.Generator(device='cpu').manual_seed(1234 + rank)
        H, W = input_size
        self.batches = []
        for _ in range(distinct):
            org = t.randn((batch_size, 3, 2 * H, 2 * W), generator=g).to(device).contiguous(memory_format=t.channels_last)
            tgt = t.randint(0, num_classes, (batch_size, 2 * H, 2 * W), generator=g, dtype=t.uint8)
            tgt[t.rand(tgt.shape, generator=g) < 0.1] = ignore
            img = HF.upsample_bilinear_ac(org, (H, W))
            self.batches.append(((img, org), (tgt.to(device), None)))
        self.length = length

    def __len__(self):
        return self.length

    def __iter__(self):
        for i in range(self.length):
            yield self.batches[i % len(self.batches)]


class _CapturedStep:
    __slots__ = ('graph', 'graph_b', 'ready', 'img', 'org', 'tgt', 'outs', 'vals', 'bns', 'keep')


class TrainStep:
    """One iteration of train_or_resume.py:404-460 for a fixed model/arena.

    With `graph` (default: on, DSRL_GRAPH=0 turns it off) a training iteration is captured once per batch shape into a hipGraph -
    key advance, gradient zeroing, filter transposes, forward, losses, backward, SGD update - and replayed: ~1000 kernel launches
    become one hipGraphLaunch, so the host no longer paces the device.  What makes the capture replayable: the dropout key and the
    optimiser hyper-parameters are read from device memory (functional.DeviceRng, dsrl_sgd_step_dev), the fused BatchNorm barrier
    is self-resetting, and the batch is copied into static input buffers.  With more than one rank the collectives stay outside the
    graph: BN-buffer broadcast before the replay, chunked gradient all-reduce and the SGD kernel after it."""

    GRAPH_WARMUP = 2          # eager iterations per batch shape before the capture (lazy initialisation, allocator warm-up)

    def __init__(self, model, flat, stage, w1, w2, ignore_index, graph=None):
        self.model, self.flat, self.stage, self.w1, self.w2, self.ignore = model, flat, stage, w1, w2, ignore_index
        self.fa = FALoss()
        dev = flat.device
        self.flag = t.zeros(1, dtype=t.int32, device=dev)
        self.zero = t.zeros((), device=dev)
        self._pending, self._free = [], []           # (pinned host buffer, event) of iterations in flight / reusable
        self.use_graph = (os.environ.get('DSRL_GRAPH', '1') != '0') if graph is None else bool(graph)
        self.fused_losses = os.environ.get('DSRL_FUSED_LOSSES', '1') != '0'
        self._graphs, self._warm = {}, {}
        self.rng = self.hyper = self._hyper_host = self._hyper_vals = None
        self.host_enqueue_s = 0.0
        self.graph_replays = 0
        if self.use_graph and flat.world > 1:
            # the collectives of a replayed step are launched between / behind its graphs (no hook-launched all-reduce).  The fused BatchNorm
            # kernels keep the 128-block budget ddp.FlatParams selected: an all-reduce of the first gradient chunks runs beside the second
            # half of backward (split capture), and RCCL blocks hold CUs while they wait for peers - a 256-block barrier launch needs every CU
            flat.defer_collectives = True
        # Two-phase backward with more than one rank (DSRL_GRAPH_SPLIT=1; off by default since round 4): backward stops at the layer3 / layer4
        # boundary, the gradient chunks complete by then (head, ASPP, layer4) are all-reduced while layers 3..1 run, the rest behind them.  That
        # schedule puts RCCL kernels beside the device-wide-barrier BatchNorm kernels of the second graph; it is bit-identical to the one-graph
        # schedule and was rehearsed on one GPU, but it has never run with two GPUs on RCCL, so it stays opt-in until a multi-GPU run has recorded
        # `bn_fused_barrier_timeouts` and the exposed all-reduce time with it.  Default: ONE graph, one all-reduce of the whole
        # 239.5 MB arena behind it (round 2's schedule).
        self.split = flat.world > 1 and flat.defer_collectives and os.environ.get('DSRL_GRAPH_SPLIT', '0') != '0'
        self.time_collectives = False           # bench.py: bracket the exposed part of the collectives with events
        self.comm_events = []                   # (broadcast start, end, replay end, end of exchange [+ update: one-graph schedule]) per step

    def losses(self, outs, input_org, target):
        SSSR, SISR, SSSR_ft, SISR_ft = outs
        ce = HF.cross_entropy(SSSR, target, self.ignore)                                   # train_or_resume.py:435
        ms = self.w1 * HF.mse_loss(SISR, input_org) if self.stage > 1 else self.zero       # :436
        fa = self.w2 * self.fa(SSSR_ft, SISR_ft) if self.stage > 2 else self.zero          # :437
        return ce, ms, fa, ce + ms + fa                                                    # :438

    # ------------------------------------------------------------------ the iteration itself (eager, or under capture)
    def _phase_a(self, input_image, input_org, target, do_train, in_graph, split):
        """Step start, forward, losses and - when training - backward: all of it, or (split) down to the layer3 / layer4 cut."""
        flat = self.flat
        bb = None
        if do_train:
            if self.rng is not None:
                self.rng.advance()                # this step's dropout key, derived on the device
            flat.zero_grad()                                                               # optimizer.zero_grad(), :418
            if not in_graph:
                flat.sync_buffers()
            flat.refresh_transposed_filters()     # one launch: the [C][R][S][K] filter copies every dgrad of this step reads
            if split:
                bb = self.model.feature_extractor['backbone']
                bb._dsrl_cut = []
        self.flag.zero_()
        cuts = None
        try:
            with t.set_grad_enabled(do_train):
                if self.fused_losses and do_train:
                    # the layer that produces the logits evaluates the CE value in its own forward kernel (HF.logits_target), :435
                    with HF.logits_target(target if target.dtype == t.uint8 and target.is_contiguous() else None, self.ignore, self.flag):
                        outs = self.model(input_image)                                     # :420
                else:
                    outs = self.model(input_image)                                         # :420
                if self.fused_losses:
                    # CE + MSE + FA, their gradients, the NaN asserts (:426-433) and the loss mix (:435-438) in one launch set
                    vals = HF.fused_losses(outs, target, input_org, self.ignore, self.w1, self.w2, self.stage, self.flag, self.fa.subsample_factor)
                    total = vals[3]
                else:
                    HF.nan_check_(self.flag, *[o for o in outs if o.is_cuda])              # the four NaN asserts, :426-433
                    ce, ms, fa, total = self.losses(outs, input_org, target)
                    vals = t.cat([t.stack([ce, ms, fa, total]).detach().float(), self.flag.float()])
                if in_graph and os.environ.get('DSRL_GRAPH_FAIL_TEST'):
                    raise RuntimeError('DSRL_GRAPH_FAIL_TEST: simulated failure in the middle of a capture')      # tests/test_rccl_gpu.py
                if do_train:
                    if self.fused_losses:
                        HF.fused_losses_backward(vals)                                     # :444 (total.backward(), the root gradient a cached constant)
                    else:
                        total.backward()                                                   # :444
                    if split:
                        cuts = bb._dsrl_cut
                        HF.flush_wgrad_queue(reopen=True)      # the weight gradients of head, ASPP and layer4 as one grouped launch set
                        HF.join_side_streams()
        finally:
            if bb is not None:
                bb._dsrl_cut = None
        return outs, vals.detach(), cuts

    def _phase_b(self, cuts):
        """Second backward phase: from the cut tensors through layers 3..1 and the stem, then their grouped weight gradients."""
        pairs = [(o, d.grad) for o, d in cuts if d.grad is not None]
        if pairs:
            t.autograd.backward([o for o, _ in pairs], [g for _, g in pairs])
        HF.flush_wgrad_queue()
        HF.join_side_streams()

    def _finish(self, hp, in_graph):
        """Gradient exchange (when it did not run between the phases) and the optimiser step."""
        flat = self.flat
        flat.settle_grads()                       # every gradient kernel of the pass has been enqueued or recorded: the arena fill could be skipped again
        hyper = self.hyper if self.rng is not None else None
        if flat.world > 1 and flat.defer_collectives:
            HF.flush_wgrad_queue()
            HF.join_side_streams()
            if in_graph:
                return                            # the collectives and the update follow the replay (_replay)
            if self.split:
                flat.reduce_rest()
                flat.sgd_step(hp[0], hp[1], hp[2], hyper=hyper, reduce=False)
            else:
                flat.reduce_chunked_and_step(hyper=hyper, hp=hp)      # the all-reduce in a few ranges, the update of each range under the next one's exchange
        else:
            flat.sgd_step(hp[0], hp[1], hp[2], hyper=hyper)                                # :445 (eager, world > 1: hook-launched chunk all-reduces overlap)

    def _body(self, input_image, input_org, target, hp, do_train, in_graph=False):
        """One whole iteration, eagerly (graph mode: the iterations before the capture, with the same order of events)."""
        split = self.split and do_train
        try:
            outs, vals, cuts = self._phase_a(input_image, input_org, target, do_train, in_graph, split)
            if do_train:
                if split:
                    self.flat.reduce_chunks(self.flat.ready_chunks())      # runs beside the second phase
                    self._phase_b(cuts)
                self._finish(hp, in_graph)
        except BaseException:
            # a step that aborts between zero_grad() and sgd_step() (an exception in forward / backward, an allocation failure) must not leave the step
            # arena open: the validation forwards that follow would draw records from it, and an arena a graph has pinned cannot grow
            if do_train and self.flat.device.type == 'cuda':
                HF.amax_end_step(self.flat.device)
                HF.wgrad_queue = None
            raise
        return outs, vals

    def _set_hyper(self, hp):
        vals = (float(hp[0]), float(hp[1]), float(hp[2]), 1.0 / self.flat.world)
        if vals != self._hyper_vals:
            self._hyper_host.copy_(t.tensor(vals, dtype=t.float32))
            self.hyper.copy_(self._hyper_host, non_blocking=True)
            self._hyper_vals = vals

    def _graph_key(self, input_image, input_org, target):
        return (tuple(input_image.shape), tuple(input_org.shape), tuple(target.shape), HF.get_conv_precision(), HF.overlap_wgrad)

    def _capture(self, key, input_image, input_org, target, hp):
        c = _CapturedStep()
        c.img, c.org, c.tgt = input_image.clone(), input_org.clone(), target.clone()
        bns = [m for m in self.model.modules() if isinstance(m, t.nn.modules.batchnorm._BatchNorm)]
        before = [getattr(m, '_dsrl_batches', 0) for m in bns]
        step_before = HF._rng_state['step']
        dump_dir = os.environ.get('DSRL_GRAPH_DEBUG_DUMP')          # tools/graph_memset_edges.py: keep the captured hipGraph_t and write it as a DOT file
        c.graph = t.cuda.CUDAGraph(keep_graph=True) if dump_dir else t.cuda.CUDAGraph()
        # A forked capture (weight gradients on a side stream) replays slower than a linear one on this runtime: the graph executor
        # pays more for its cross-queue dependencies than the overlap wins (measured 24.2 vs 23.8 ms per step), so the capture is
        # linear unless DSRL_GRAPH_OVERLAP=1
        overlap_was = HF.overlap_wgrad
        fused_was = None
        if os.environ.get('DSRL_GRAPH_OVERLAP', '0') == '0':
            HF.overlap_wgrad = False
        else:
            # a forked capture can place fused BatchNorm nodes on parallel branches: they share ONE device-wide barrier word pair and must
            # never overlap each other, so that capture takes the three-kernel BatchNorm path
            fused_was = HF.set_bn_fused_max_blocks(0)
        c.keep = HF.graph_keepalive = []          # pinned host tables the captured copies read on every replay
        HF.capture_host, HF.capture_host_off = t.empty(2 << 20, dtype=t.uint8, pin_memory=True), 0      # allocated BEFORE the capture starts
        c.keep.append(HF.capture_host)
        if HF.f16_mode():
            c.keep.append(HF.amax_pin(self.flat.device))      # the graph zeroes, maxes into and reads this arena on every replay: it lives as long as the graph
        c.graph_b, c.ready = None, []
        mode = os.environ.get('DSRL_GRAPH_CAPTURE_MODE', 'thread_local')
        try:
            if self.split:
                # two graphs sharing one memory pool: [step start .. backward down to the cut + its weight gradients] and [the rest of backward];
                # the chunks complete after the first are recorded - on replay their all-reduce is launched between the two graphs
                with t.cuda.graph(c.graph, capture_error_mode=mode):
                    c.outs, c.vals, cuts = self._phase_a(c.img, c.org, c.tgt, True, True, True)
                c.ready = self.flat.ready_chunks()
                c.graph_b = t.cuda.CUDAGraph(keep_graph=True) if dump_dir else t.cuda.CUDAGraph()
                with t.cuda.graph(c.graph_b, pool=c.graph.pool(), capture_error_mode=mode):
                    self._phase_b(cuts)
                del cuts
            else:
                with t.cuda.graph(c.graph, capture_error_mode=mode):
                    c.outs, c.vals, _ = self._phase_a(c.img, c.org, c.tgt, True, True, False)
                    self._finish(hp, True)
        finally:
            HF.overlap_wgrad = overlap_was
            if fused_was is not None:
                HF.set_bn_fused_max_blocks(fused_was)
            HF.graph_keepalive = None
            HF.capture_host = None
        if dump_dir:
            import ctypes
            os.makedirs(dump_dir, exist_ok=True)
            hip = ctypes.CDLL('libamdhip64.so')
            hip.hipGraphDebugDotPrint.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint]
            for g_, name in ((c.graph, 'graph_a.dot'), (c.graph_b, 'graph_b.dot')):
                if g_ is not None:
                    rc = hip.hipGraphDebugDotPrint(ctypes.c_void_p(g_.raw_cuda_graph()), os.path.join(dump_dir, name).encode(), 1)      # 1 = verbose
                    print(f'[dsrl] hipGraphDebugDotPrint({name}) -> {rc}', flush=True)
                    g_.instantiate()
        # nothing ran during the capture: take back the host-side bookkeeping of that phantom iteration
        HF._rng_state['step'] = step_before
        c.bns = [m for m, n in zip(bns, before) if getattr(m, '_dsrl_batches', 0) != n]
        for m, n in zip(bns, before):
            if hasattr(m, '_dsrl_batches'):
                m._dsrl_batches = n
        self._graphs[key] = c
        return c

    def _capture_or_fall_back(self, key, input_image, input_org, target, hp):
        """A capture that fails (a runtime that refuses something inside it) must not end the training run: this TrainStep then keeps
        launching eagerly - the same kernels and, with more than one rank, the same collectives in the same order, so ranks that captured and
        ranks that did not stay in step.  Nothing of a failed capture has executed; its host-side bookkeeping is rolled back."""
        import sys
        bns = [m for m in self.model.modules() if isinstance(m, t.nn.modules.batchnorm._BatchNorm)]
        before = [getattr(m, '_dsrl_batches', 0) for m in bns]
        step_before, overlap_was = HF._rng_state['step'], HF.overlap_wgrad
        try:
            return self._capture(key, input_image, input_org, target, hp)
        except Exception as e:          # noqa: BLE001
            print(f'[dsrl] hipGraph capture failed ({type(e).__name__}: {str(e)[:200]}); this TrainStep continues with eager launches', file=sys.stderr, flush=True)
            self.use_graph = False
            HF.overlap_wgrad, HF.graph_keepalive, HF.capture_host, HF.wgrad_queue = overlap_was, None, None, None
            HF._rng_state['step'] = step_before
            for m, n in zip(bns, before):
                if hasattr(m, '_dsrl_batches'):
                    m._dsrl_batches = n
            try:
                t.cuda.set_stream(t.cuda.default_stream(self.flat.device))
                t.cuda.synchronize(self.flat.device)
            except Exception:           # noqa: BLE001
                pass
            return None

    def _replay(self, c, input_image, input_org, target):
        flat = self.flat
        for dst, src in ((c.img, input_image), (c.org, input_org), (c.tgt, target)):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        ev = [t.cuda.Event(enable_timing=True) for _ in range(4)] if (self.time_collectives and flat.world > 1) else None
        if flat.world > 1:
            if ev:
                ev[0].record()
            flat.sync_buffers()
            if ev:
                ev[1].record()
            flat._reduced, flat._works = set(), []      # no hook fires during a replay: the chunk bookkeeping of the step starts here
        flat.ensure_filter_amax()           # the graph may rely on the filter magnitudes the previous optimiser pass left (ddp.FlatParams._sgd_range)
        c.graph.replay()
        if c.graph_b is not None:
            flat.reduce_chunks(c.ready)         # head / ASPP / layer4 chunks: their all-reduce runs beside the second graph
            c.graph_b.replay()
        HF._rng_state['step'] += 1              # host mirror of the key the device just derived
        HF._rng_state['current'] = HF._derive(HF._rng_state['step'])
        for m in c.bns:
            m._dsrl_batches += 1
        if flat.world > 1:
            if ev:
                ev[2].record()
            if c.graph_b is not None:
                flat.reduce_rest()
                if ev:
                    ev[3].record()
                flat.sgd_step(0.0, 0.0, 0.0, hyper=self.hyper, reduce=False)
            else:
                flat.reduce_chunked_and_step(hyper=self.hyper)       # exchange + update: SGD of range k under the all-reduce of range k + 1
                if ev:
                    ev[3].record()
            if ev:
                self.comm_events.append(ev)
        self.graph_replays += 1
        return c.outs, c.vals

    def enqueue(self, input_image, input_org, target, lr, momentum, weight_decay, do_train=True):
        """Enqueue one whole iteration on the device and start the asynchronous device->host copy of its five scalars (CE, MSE,
        FA, total, NaN flag) into pinned memory.  Nothing is waited for: collect() returns the values of the OLDEST iteration
        still outstanding.  Calling enqueue(k+1) before collect(k) keeps the launch queue fed across the iteration boundary
        (the reference reads its losses synchronously every iteration, train_or_resume.py:457-460; here the read of
        iteration k overlaps the enqueue of k+1 and the NaN assert fires one iteration late).  Returns the network outputs
        (in graph mode: the capture's static output tensors, overwritten by the next replay)."""
        import time
        t_host0 = time.perf_counter()
        hp = (lr, momentum, weight_decay)
        if do_train and self.use_graph:
            if self.rng is None:
                self.rng = HF.DeviceRng(self.flat.device)
                self.hyper = t.zeros(4, device=self.flat.device)
                self._hyper_host = t.empty(4, dtype=t.float32, pin_memory=True)
            self._set_hyper(hp)
            key = self._graph_key(input_image, input_org, target)
            c = self._graphs.get(key)
            if c is None and self._warm.get(key, 0) >= self.GRAPH_WARMUP:
                c = self._capture_or_fall_back(key, input_image, input_org, target, hp)
            if c is None:
                self._warm[key] = self._warm.get(key, 0) + 1
                outs, vals = self._body(input_image, input_org, target, hp, True)
            else:
                outs, vals = self._replay(c, input_image, input_org, target)
        else:
            outs, vals = self._body(input_image, input_org, target, hp, do_train)
        if len(self._free) == 0:
            self._free.append((t.empty(5, dtype=t.float32, pin_memory=True), t.cuda.Event()))
        host, ev = self._free.pop()
        host.copy_(vals, non_blocking=True)                                                # ONE device->host read per iteration
        ev.record()
        self._pending.append((host, ev))
        self.host_enqueue_s = time.perf_counter() - t_host0                   # time the host needed to enqueue the whole step
        return outs

    def pending(self):
        return len(self._pending)

    def collect(self):
        """Losses [CE, MSE, FA, total] of the oldest outstanding iteration (waits for it); raises on NaN network outputs."""
        host, ev = self._pending.pop(0)
        ev.synchronize()
        vals = [float(v) for v in host]
        self._free.append((host, ev))
        if int(vals[4]) & 2:
            raise AssertionError('a target label is outside [0, num_classes) and is not the ignore index (CrossEntropyLoss would assert).')
        if vals[4] != 0:
            stuck = HF.bn_fused_barrier_timeouts()
            if stuck:
                raise RuntimeError(f'{stuck} blocks of the fused BatchNorm kernels timed out at their device-wide barrier and poisoned their outputs: '
                                   'is another process using this GPU? (functional.set_bn_fused_max_blocks(0) selects the three-kernel path)')
            raise AssertionError("network output contains 'NaN' values and so cannot continue.")
        return vals[:4]

    def release(self):
        """Drops the captured graphs and unbinds the device-resident dropout key (launches take their `seed` argument again)."""
        self._graphs.clear()
        if self.rng is not None:
            self.rng.release()
            self.rng = None

    def __call__(self, input_image, input_org, target, lr, momentum, weight_decay, do_train=True):
        """Synchronous form: enqueue + collect of the same iteration."""
        while self._pending:
            self.collect()
        outs = self.enqueue(input_image, input_org, target, lr, momentum, weight_decay, do_train)
        return self.collect(), outs


def _get_state_dict(model):
    return model.state_dict()


def train_or_resume(is_resuming_training, device, distributed, mixed_precision, disable_cudnn_benchmark, num_workers, dataset, val_interval,
                    checkpoint_interval, checkpoint_history, init_weights, batch_size, epochs, learning_rate, end_learning_rate, momentum,
                    weights_decay, poly_power, stage, w1, w2, freeze_batch_norm, experiment_id, description, early_stopping, dry_run=False, **other_args):
    if not isCUDAdevice(device):
        raise RuntimeError("this build runs on the MI355X only: use device='gpu' (the reference's --device cpu path is its own)")
    if mixed_precision not in settings.MIXED_PRECISION_TO_CONV_ARITHMETIC:
        raise RuntimeError(f"mixed_precision={mixed_precision!r}: expected one of {sorted(k for k in settings.MIXED_PRECISION_TO_CONV_ARITHMETIC if k)} "
                           '(apex itself is not used: the opt level selects the MFMA conv arithmetic)')
    conv_arith = settings.MIXED_PRECISION_TO_CONV_ARITHMETIC[mixed_precision]
    input_size = other_args.get('model_input_size', settings.MODEL_INPUT_SIZE)
    if distributed:
        t.manual_seed(settings.RANDOM_SEED)                                                # identical init on all ranks, :31
        if not dist.is_initialized():
            dist.init_process_group(distributed['BACKEND'], distributed['INIT_METHOD'], world_size=distributed['WORLD_SIZE'], rank=distributed['RANK'])
        if not dist.is_initialized():
            raise RuntimeError("Couldn't initialize distributed process group!")
        is_master_rank = (distributed['RANK'] == 0)
        device_obj = t.device('cuda', distributed['DEVICE_ID'])
        rank, world = distributed['RANK'], distributed['WORLD_SIZE']
    else:
        is_master_rank, device_obj, rank, world = True, t.device('cuda', t.cuda.current_device()), 0, 1
    t.cuda.set_device(device_obj)
    if is_master_rank:
        process_start_timestamp = datetime.now()
        best_validation_dict = other_args['best_validation_dict'] if is_resuming_training else {'epoch': -1, 'best_miou_percent': 0., 'loss': 0.}

    ds = dataset['settings']
    model = DSRL(stage, ds)                                                                # :62
    if is_resuming_training:
        model.load_state_dict(other_args['model_state_dict'], strict=True)
        starting_epoch = other_args['epoch']
    else:
        starting_epoch = 0
        if init_weights:
            model.load_state_dict(t.load(init_weights, map_location='cpu')['model_state_dict'], strict=False)
        elif stage == 1:
            if other_args.get('pretrained_backbone', True):
                model.initialize_with_pretrained_weights(settings.WEIGHTS_ROOT_DIR)
        else:
            prev = os.path.join(experiment_id, settings.WEIGHTS_DIR.format(stage=stage - 1), settings.FINAL_WEIGHTS_FILE)
            if os.path.isfile(prev):
                model.load_state_dict(t.load(prev, map_location='cpu')['model_state_dict'], strict=False)      # :91-96
            elif other_args.get('pretrained_backbone', True):
                model.initialize_with_pretrained_weights(settings.WEIGHTS_ROOT_DIR)
    model = model.to(device_obj).to(memory_format=t.channels_last)                        # :103 (+ kernel weight layout)
    flat = FlatParams(model)                                                               # DDP wrap, :105-106
    if is_resuming_training and 'optimizer_state_dict' in other_args:
        flat.load_state_dict(other_args['optimizer_state_dict'])
    step = TrainStep(model, flat, stage, w1, w2, ds.IGNORE_CLASS_LABEL)

    factory = dataset.get('loader_factory')
    if factory is None:
        os.makedirs(dataset['path'], exist_ok=True)
        if len(os.listdir(dataset['path'])) == 0:
            raise Exception("Cityscapes dataset was not found under '{:s}'.".format(dataset['path']))
        raise NotImplementedError("the torchvision/PIL input pipeline is out of scope here: pass dataset['loader_factory']")
    train_loader = factory('train', batch_size, device_obj, rank, world)
    val_loader = factory('val', batch_size, device_obj, rank, world) if is_master_rank else None

    history = []
    CE_val_avg_loss = MSE_val_avg_loss = FA_val_avg_loss = Avg_val_loss = None
    amp_state_dict = None

    def save_checkpoint(filename, epoch, train_means):
        """utils.py:273-275 with every key of settings.VARIABLES_IN_CHECKPOINT (train_or_resume.py:268-282, 318-332)."""
        ckpt_dir = os.path.join(experiment_id, settings.CHECKPOINTS_DIR.format(stage=stage))
        os.makedirs(ckpt_dir, exist_ok=True)
        have = {'device': device, 'mixed_precision': mixed_precision, 'amp_state_dict': amp_state_dict, 'disable_cudnn_benchmark': disable_cudnn_benchmark,
                'num_workers': num_workers, 'val_interval': val_interval, 'checkpoint_interval': checkpoint_interval, 'checkpoint_history': checkpoint_history,
                'init_weights': init_weights, 'batch_size': batch_size, 'epochs': epochs, 'learning_rate': learning_rate,
                'end_learning_rate': end_learning_rate, 'momentum': momentum, 'weights_decay': weights_decay, 'poly_power': poly_power, 'stage': stage,
                'w1': w1, 'w2': w2, 'freeze_batch_norm': freeze_batch_norm, 'experiment_id': experiment_id, 'description': description,
                'early_stopping': early_stopping, 'CE_train_avg_loss': train_means[0], 'MSE_train_avg_loss': train_means[1],
                'FA_train_avg_loss': train_means[2], 'Avg_train_loss': train_means[3], 'CE_val_avg_loss': CE_val_avg_loss,
                'MSE_val_avg_loss': MSE_val_avg_loss, 'FA_val_avg_loss': FA_val_avg_loss, 'Avg_val_loss': Avg_val_loss, 'epoch': epoch,
                'best_validation_dict': best_validation_dict, 'model_state_dict': _get_state_dict(model),
                'optimizer_state_dict': flat.state_dict(lr, momentum, weights_decay, initial_lr=learning_rate)}
        t.save({k: have[k] for k in settings.VARIABLES_IN_CHECKPOINT}, os.path.join(ckpt_dir, filename))
        return ckpt_dir

    if conv_arith is not None:
        HF.set_conv_precision(conv_arith)
    try:
        for epoch in range(starting_epoch + 1, epochs + 1):
            lr = polynomial_lr(learning_rate, end_learning_rate, epoch - 1, epochs, poly_power)       # scheduler.step() per epoch, :349
            means = _do_train_val(True, epoch, model, step, train_loader, lr, momentum, weights_decay, freeze_batch_norm, is_master_rank)
            rec = {'epoch': epoch, 'lr': lr, 'train': means}
            stop = False
            if is_master_rank:
                if checkpoint_history > 0 and epoch % checkpoint_interval == 0 and experiment_id:                 # :264
                    CE_val_avg_loss = MSE_val_avg_loss = FA_val_avg_loss = Avg_val_loss = None                      # :270-273
                    ckpt_dir = save_checkpoint(settings.CHECKPOINT_FILE.format(epoch=epoch), epoch, means)
                    old = epoch - checkpoint_history * checkpoint_interval                                            # :285-291
                    if old > 0:
                        stale = os.path.join(ckpt_dir, settings.CHECKPOINT_FILE.format(epoch=old))
                        if os.path.isfile(stale):
                            os.remove(stale)
                if val_loader is not None and epoch % val_interval == 0:                                            # :294
                    rec['val'] = _do_train_val(False, epoch, model, step, val_loader, lr, momentum, weights_decay, False, True)
                    CE_val_avg_loss, MSE_val_avg_loss, FA_val_avg_loss, Avg_val_loss, val_mIoU = rec['val'][:5]
                    if val_mIoU > best_validation_dict['best_miou_percent']:                                        # :318-337
                        best_validation_dict = {'epoch': epoch, 'best_miou_percent': val_mIoU, 'loss': Avg_val_loss}
                        if experiment_id:
                            ckpt_dir = os.path.join(experiment_id, settings.CHECKPOINTS_DIR.format(stage=stage))
                            for x in glob.glob(os.path.join(ckpt_dir, '*_bestval.checkpoint')):
                                if os.path.isfile(x):
                                    os.remove(x)
                            save_checkpoint(settings.CHECKPOINT_FILE.format(epoch='{:d}_bestval'.format(epoch)), epoch, means)
                    if means[3] < Avg_val_loss and early_stopping:                                                  # :341-347
                        rec['early_stopped'] = stop = True
            history.append(rec)
            if distributed:
                # the reference breaks on the master rank only (its other ranks would hang in the next collective); here every rank learns it
                flag = t.tensor([1 if stop else 0], device=device_obj)
                dist.broadcast(flag, 0)
                stop = bool(flag.item())
            if stop:
                break
    finally:
        step.release()
        if conv_arith is not None:
            HF.set_conv_precision(None)
    if is_master_rank and experiment_id:
        wdir = os.path.join(experiment_id, settings.WEIGHTS_DIR.format(stage=stage))
        os.makedirs(wdir, exist_ok=True)
        t.save({'model_state_dict': _get_state_dict(model), 'mixed_precision': mixed_precision, 'amp_state_dict': amp_state_dict},
               os.path.join(wdir, settings.FINAL_WEIGHTS_FILE))                               # utils.py:277-282
        history.append({'elapsed': str(datetime.now() - process_start_timestamp)})
    return history


def _do_train_val(do_train, epoch, model, step, data_loader, lr, momentum, weights_decay, freeze_batch_norm, is_master_rank):
    """train_or_resume.py:373-533 without the progress-bar / TensorBoard plumbing. Returns the running means
    (CE, MSE, FA, Total, mIoU %, accuracy %); the last two only for validation."""
    model.train(mode=do_train)
    if do_train and freeze_batch_norm:
        for m in model.modules():
            if isinstance(m, t.nn.modules.batchnorm._BatchNorm):
                m.eval()                                                                   # :379-382
    from ..metrices import Accuracy, AverageMeter, mIoU
    meters = [AverageMeter() for _ in range(4)]                                            # CE, MSE, FA, Total (:385-388)
    nc = model.SSSR_decoder['cls_conv'].out_channels
    miou, mean_accuracy = mIoU(num_classes=nc, ignore_index=step.ignore), Accuracy(num_classes=nc, ignore_index=step.ignore)
    sizes = []

    def drain(keep):
        while step.pending() > keep:
            vals, n = step.collect(), sizes.pop(0)
            for mtr, v in zip(meters, vals):
                mtr.update(v, n)                                                           # AverageMeter.update(value, batch), :457-460

    for (input_image, input_org), (target, _) in data_loader:
        outs = step.enqueue(input_image, input_org, target, lr, momentum, weights_decay, do_train)
        sizes.append(input_image.shape[0])
        if not do_train and is_master_rank:
            miou.update_from_logits(outs[0], target)                                       # argmax + histograms on the device (:476-480)
            mean_accuracy.update_from_logits(outs[0], target)
        drain(1)              # the losses of iteration k are read while iteration k+1 is already queued
    drain(0)
    return [m() for m in meters] + [miou() if not do_train else 0.0, mean_accuracy() if not do_train else 0.0]

