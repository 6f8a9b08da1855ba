"""Data-parallel gradient reduction and the SGD update on flat fp32 arenas.

Replaces what the reference gets from `torch.nn.parallel.DistributedDataParallel(model)` + `torch.optim.SGD`
(command_handlers/train_or_resume.py:63-66, 105-106, 444-445) with a layout chosen for RCCL over xGMI:

  * all parameters live in ONE contiguous arena (and their gradients / momentum buffers in two more), ordered in reverse
    registration order, i.e. roughly the order in which backward produces gradients;
  * the gradient arena is all-reduced in a few large contiguous chunks (default 32 MiB) straight out of that memory - no
    bucket gather/scatter copies - and each chunk is launched asynchronously from an autograd hook as soon as every
    gradient in it has been accumulated, so RCCL overlaps with the rest of backward;
  * the mean (1/world) is folded into the fused SGD kernel (one launch over the whole arena);
  * BatchNorm running statistics sit in a fourth small arena that rank 0 broadcasts in one collective per step (DDP's
    broadcast_buffers=True default, which train_or_resume.py:106 relies on).

One process per GPU; `torch.distributed` backend 'nccl' is RCCL on ROCm.  With world size 1 (or no process group) the
collectives are skipped and only the arena + fused optimiser remain.
"""
import os

import torch
import torch.distributed as dist

from . import functional as HF


def _align(n, a=4):
    return (n + a - 1) // a * a


class FlatParams:
    def __init__(self, model, chunk_bytes=32 << 20, process_group=None, broadcast_buffers=True):
        self.model = model
        params = [p for p in model.parameters() if p.requires_grad]
        if not params:
            raise ValueError('model has no trainable parameters')
        dev = params[0].device
        self.device = dev
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.params = list(reversed(params))                 # backward order first
        offs, off = [], 0
        for p in self.params:
            offs.append(off)
            off += _align(p.numel())
        self.offsets, self.numel = offs, off
        self.p_flat = torch.zeros(off, device=dev, dtype=torch.float32)
        self.g_flat = torch.zeros(off, device=dev, dtype=torch.float32)
        self.m_flat = torch.zeros(off, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for p, o in zip(self.params, offs):
                if p.dtype != torch.float32:
                    raise TypeError('fp32 parameters expected')
                if not (p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))):
                    p.data = p.data.contiguous()          # the arena views need dense storage
                view = self.p_flat.as_strided(p.shape, p.stride(), o)
                view.copy_(p.data)
                p.data = view
                p.grad = self.g_flat.as_strided(p.shape, p.stride(), o)
        # float buffers (BN running statistics) in their own arena
        self.buffers = []
        boff = 0
        for mod in model.modules():
            for name, b in list(mod._buffers.items()):
                if b is not None and b.dtype == torch.float32:
                    self.buffers.append((mod, name, boff, b))
                    boff += _align(b.numel())
        self.b_flat = torch.zeros(max(boff, 4), device=dev, dtype=torch.float32)
        with torch.no_grad():
            for mod, name, o, b in self.buffers:
                view = self.b_flat[o:o + b.numel()].view(b.shape)
                view.copy_(b)
                mod._buffers[name] = view
        self.broadcast_buffers_enabled = broadcast_buffers
        # chunks of the gradient arena, each a contiguous range of whole parameters
        per = max(1, chunk_bytes // 4)
        self.chunks, start, count = [], 0, 0
        self._chunk_of = {}
        for i, (p, o) in enumerate(zip(self.params, offs)):
            self._chunk_of[i] = len(self.chunks)
            count += 1
            end = o + _align(p.numel())
            if end - start >= per or i == len(self.params) - 1:
                self.chunks.append([start, end, count])
                start, count = end, 0
        self._pending = [c[2] for c in self.chunks]
        self._works = []
        self._reduced = set()
        self._hooks = []
        self.defer_collectives = False      # set while a step is captured into a hipGraph: reduce_all() runs the collectives after the replay
        # gradient sink: kernels may write a parameter's gradient straight into its arena slot (functional._sink)
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self._claimed = set()
        # Round 5: zero_grad() skips the fill of the gradient arena (239.5 MB, 31 us) when every parameter's gradient kernel OVERWROTE its slot in the
        # previous pass (claim: all 354 parameters of the stage-3 step do) - the same kernels will overwrite them again.  settle_grads() checks the
        # assumption behind the skipped fill after the backward pass and refuses to go on if it did not hold (DSRL_LAZY_ZERO_GRAD=0: always fill).
        self.lazy_zero = os.environ.get('DSRL_LAZY_ZERO_GRAD', '1') != '0'
        self._all_claimed_last = False
        self._fill_skipped = False
        for p in self.params:
            p._dsrl_arena = self
        self._build_transposed_filters()
        if self.world > 1 and self.device.type == 'cuda' and os.environ.get('DSRL_BN_FUSED_BIG') is None:
            # RCCL kernels hold CUs while they wait for their peers: a 256-block fused-BN launch (one block on EVERY CU) could stall behind
            # them, the 128-block variant cannot
            HF.set_bn_fused_max_blocks(128)
        if self.world > 1:
            dist.broadcast(self.p_flat, 0, group=self.pg)          # DDP constructor semantics: rank 0's weights win
            dist.broadcast(self.b_flat, 0, group=self.pg)
            for i, p in enumerate(self.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))

    # ------------------------------------------------------------------ gradient reduction
    def _make_hook(self, i):
        def hook(_param):
            # torch runs the post-accumulate hook of a parameter even when its backward returned no gradient, i.e. also for a parameter
            # whose kernel writes straight into the arena (claim): that gradient is complete when the kernel has been ENQUEUED - for a
            # deferred weight gradient at the flush of the queue, long after this hook - and is reported by written() alone.  (Counting both
            # made a chunk look complete after half of its notifications: found by the two-phase reduction, round 3.)
            if i in self._claimed:
                return
            self._on_grad_ready(i)
        return hook

    # ------------------------------------------------------------------ gradient sink protocol
    def claim(self, p):
        """First gradient of `p` in this backward pass? Then the producing kernel may overwrite the (zeroed) arena slot."""
        i = self._index.get(id(p))
        if i is None or i in self._claimed or p.grad is None or p.grad.data_ptr() != self.g_flat.data_ptr() + 4 * self.offsets[i]:
            return False
        self._claimed.add(i)
        return True

    def written(self, p, stream=None):
        """The kernel that claimed `p` has been enqueued (on `stream`, default the current one): run the reducer hook autograd
        would have run. Gradients produced on a side stream are joined before the collective is launched."""
        if self.world > 1:
            self._on_grad_ready(self._index[id(p)])

    def _on_grad_ready(self, i):
        ci = self._chunk_of[i]
        self._pending[ci] -= 1
        if self._pending[ci] == 0 and not self.defer_collectives:
            a, b, _ = self.chunks[ci]
            if self.device.type == 'cuda' and HF.overlap_wgrad:
                # The chunk holds gradients written on the compute stream AND weight gradients written on the side stream.  Launch the
                # collective with the side stream current, after making that stream wait for the compute stream: the collective's own
                # stream then starts behind every producer of the chunk on both streams, and the compute stream itself never waits
                # (it used to join the side stream here, draining the weight-gradient backlog at every chunk boundary).
                cur = torch.cuda.current_stream(self.device)
                side = HF.side_stream(self.device)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    self._works.append(dist.all_reduce(self.g_flat[a:b], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
            else:
                self._works.append(dist.all_reduce(self.g_flat[a:b], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    # ------------------------------------------------------------------ transposed conv filters for the data-gradient kernels
    def _build_transposed_filters(self):
        """One arena with the [C][R][S][K padded to 4] transpose of every conv filter the dgrad kernel reads, refreshed by ONE
        launch per training step (dsrl_conv2d_transpose_filters_batched) instead of one launch inside every dgrad call."""
        from .nn_modules import HipConv2d
        self.wt_valid, self.wt_fp32_valid, self.split_valid, self._wt_table, self._wt_tiles = False, False, False, None, 0
        self._split_entries, self._split_table, self._seg_tables, self._amax_key = [], None, {}, None
        if self.device.type != 'cuda':
            return
        mine = {id(p) for p in self.params}
        entries, floats = [], 0
        for m in self.model.modules():
            if not isinstance(m, HipConv2d):
                continue
            w = m.weight
            K, C, R, S = w.shape
            if id(w) not in mine or C % 4 != 0 or not HF._is_krsc(w):
                continue            # the RGB stem (C = 3) and the C -> 1 feature transformers never run the implicit-GEMM dgrad
            Kp = (K + 3) & ~3
            entries.append((w, K, Kp, R * S, C, floats))
            floats += _align(C * R * S * Kp)
        if not entries:
            return
        self.wt_flat = torch.empty(floats, device=self.device, dtype=torch.float32)
        # one magnitude word per filter (max |w| as a bit pattern), re-measured by the same launch: the filter-side operand scale of the
        # f16x3 conv arithmetic (include/dsrl_hip.h: dsrl_amax)
        self.w_amax = torch.zeros(len(entries) * HF.AMAX_WORDS, device=self.device, dtype=torch.int32)      # one amax record per filter
        rows, tiles = [], 0
        for i, (w, K, Kp, RS, C, off) in enumerate(entries):
            wt = self.wt_flat[off:off + C * RS * Kp]
            w._dsrl_wt = wt
            w._dsrl_wamax = self.w_amax[i * HF.AMAX_WORDS:(i + 1) * HF.AMAX_WORDS]
            ct = (C + 31) // 32
            rows.append([w.data_ptr(), wt.data_ptr(), K, Kp, RS, C, tiles, ct, self.w_amax.data_ptr() + 4 * HF.AMAX_WORDS * i, 0])
            tiles += RS * ct * ((Kp + 31) // 32)
        self._wt_table = torch.tensor(rows, dtype=torch.int64, device=self.device)
        self._wt_rows, self._wt_tiles = len(rows), tiles
        # f16x3 arithmetic: every filter also in pre-split ("plane") form, forward layout and transposed (dsrl_conv2d_split_filters_batched);
        # the arenas are allocated on first use
        self._split_entries, self._split_table = entries, None
        self.planes_valid, self._plane_sets = False, []
        self._seg_tables, self._amax_key = {}, None        # segment tables of the optimiser pass by arena range; key under which its filter magnitudes are valid

    def _build_split_filters(self):
        floats_w = sum(_align(w.numel()) for w, *_ in self._split_entries)
        self.wsplit_flat = torch.empty(floats_w, device=self.device, dtype=torch.float32)
        self.wtsplit_flat = torch.empty(self.wt_flat.numel(), device=self.device, dtype=torch.float32)
        rows, off_w = [], 0
        base = self._wt_table.cpu()
        for i, (w, K, Kp, RS, C, off) in enumerate(self._split_entries):
            wsp = self.wsplit_flat[off_w:off_w + w.numel()]
            wtsp = self.wtsplit_flat[off:off + C * RS * Kp]
            off_w += _align(w.numel())
            w._dsrl_wsplit, w._dsrl_wtsplit = wsp, wtsp
            r = base[i].tolist()
            rows.append([r[0], wtsp.data_ptr(), K, Kp, RS, C, r[6], r[7], r[8], wsp.data_ptr()])
        self._split_table = torch.tensor(rows, dtype=torch.int64, device=self.device)
        amax_only = base.clone()
        amax_only[:, 1] = 0                     # no transposed fp32 copy
        self._amax_only_table = amax_only.to(self.device)
        # the records alone, by a streaming launch over segments of every filter's contiguous storage (dsrl_conv2d_filters_amax_batched)
        seg = int(HF.query('dsrl_conv2d_filters_amax_segment_floats'))
        segs = []
        for i, (w, K, Kp, RS, C, off) in enumerate(self._split_entries):
            n, rec = w.numel(), self.w_amax.data_ptr() + 4 * HF.AMAX_WORDS * i
            for a in range(0, n, seg):
                segs.append([w.data_ptr() + 4 * a, min(seg, n - a), rec])
        self._amax_seg_table = torch.tensor(segs, dtype=torch.int64, device=self.device)
        self._amax_segs = len(segs)

    # ------------------------------------------------------------------ optimiser pass that also measures the filters (round 5)
    def _params_key(self):
        """Changes whenever torch code writes a filter or the arena (load_state_dict, copy_, fill_ ...): the magnitudes the optimiser kernel left are
        trusted only while it is unchanged - the kernels themselves write through raw pointers and do not move it."""
        return (self.p_flat._version, sum(e[0]._version for e in self._split_entries))

    def _segments(self, a, b):
        """Device table {first float, floats, amax record address or 0} covering arena range [a, b) (multiples of 4) with segments of at most 32768
        floats, each inside one parameter; the filters of _split_entries carry their record (dsrl_sgd_step_dev_segments)."""
        key = (a, b)
        tab = self._seg_tables.get(key)
        if tab is not None:
            return tab
        seg = int(HF.query('dsrl_conv2d_filters_amax_segment_floats'))
        rec_of = {id(w): self.w_amax.data_ptr() + 4 * HF.AMAX_WORDS * i for i, (w, *_r) in enumerate(self._split_entries)}
        rows = []
        for p_, o in zip(self.params, self.offsets):
            lo, hi = max(o, a), min(o + _align(p_.numel()), b)
            rec = rec_of.get(id(p_), 0)
            x = lo
            while x < hi:
                n = min(seg, hi - x)
                rows.append([x, n, rec])
                x += n
        tab = self._seg_tables[key] = (torch.tensor(rows, dtype=torch.int64, device=self.device), len(rows))
        return tab

    def _sgd_range(self, a, b, hyper, hp):
        """SGD on arena range [a, b): with device-resident hyper-parameters and the f16 arithmetics, by the segment kernel that also measures the filters."""
        if hyper is not None and self._fold_amax():
            tab, n = self._segments(a, b)
            HF.call('dsrl_sgd_step_dev_segments', self.p_flat.data_ptr(), self.g_flat.data_ptr(), self.m_flat.data_ptr(), tab.data_ptr(), n, hyper.data_ptr(), HF._stream())
        elif hyper is not None:
            HF.sgd_step_dev_(self.p_flat[a:b], self.g_flat[a:b], self.m_flat[a:b], hyper)
        else:
            HF.sgd_step_(self.p_flat[a:b], self.g_flat[a:b], self.m_flat[a:b], hp[0], hp[1], hp[2], 1.0 / self.world)

    def ensure_filter_amax(self):
        """In front of a REPLAYED step: a graph captured while the optimiser's magnitudes were valid contains no measuring sweep and relies on the
        previous replay's optimiser pass.  If torch code wrote parameters since (a restored state, load_state_dict), measure eagerly once."""
        if self._fold_amax() and self._amax_key != self._params_key():
            self.w_amax.zero_()
            HF.call('dsrl_conv2d_filters_amax_batched', self._amax_seg_table.data_ptr(), self._amax_segs, HF._stream())
            self._amax_key = self._params_key()

    def _fold_amax(self):
        return (self.device.type == 'cuda' and self._wt_table is not None and HF.f16_mode() and os.environ.get('DSRL_PRESPLIT', '1') != '0' and
                os.environ.get('DSRL_FILTER_AMAX_STREAM', '1') != '0' and os.environ.get('DSRL_SGD_AMAX', '1') != '0' and self._split_table is not None)

    def _build_plane_filters(self):
        """The filters whose convs take plane operands (channel counts multiples of 8), as fp16 planes, forward [K][R][S][C] and transposed
        [C][R][S][K]: the filter operand of conv_planes_kernel (dsrl_conv2d_filter_planes_batched), scaled by the same amax records as the split forms.
        'auto': only the filters whose convs asked for planes in an earlier step (functional._Conv2d marks them: operands large enough for the split pass
        to pay); 'all': every eligible filter.  Sets are only ever ADDED: a captured graph keeps launching the table it was captured with and reading the
        arenas of that table (another batch shape may want more filters later - they get a set of their own; nothing a graph references is replaced)."""
        have = {id(w) for st in self._plane_sets for w in st['filters']}
        ents = [(i, e) for i, e in enumerate(self._split_entries) if e[1] % 8 == 0 and e[4] % 8 == 0 and id(e[0]) not in have and
                (HF.planes_mode == 'all' or getattr(e[0], '_dsrl_want_planes', False))]
        if not ents:
            return
        lo = lambda n: int(HF.cquery('dsrl_planes_lo_offset', n))        # noqa: E731
        total = sum(2 * lo(w.numel()) for _, (w, *_r) in ents)
        wplanes = torch.empty(total, device=self.device, dtype=torch.uint8)
        wtplanes = torch.empty(total, device=self.device, dtype=torch.uint8)
        rows, off, tiles = [], 0, 0
        for i, (w, K, Kp, RS, C, _off) in ents:
            nb = 2 * lo(w.numel())
            wp, wtp = wplanes[off:off + nb], wtplanes[off:off + nb]
            off += nb
            w._dsrl_wplanes, w._dsrl_wtplanes = wp, wtp
            ct = (C + 31) // 32
            rows.append([w.data_ptr(), wtp.data_ptr(), K, K, RS, C, tiles, ct, self.w_amax.data_ptr() + 4 * HF.AMAX_WORDS * i, wp.data_ptr()])
            tiles += RS * ct * ((K + 31) // 32)
        self._plane_sets.append({'table': torch.tensor(rows, dtype=torch.int64, device=self.device), 'rows': len(rows), 'tiles': tiles,
                                 'arenas': (wplanes, wtplanes), 'filters': [e[0] for _, e in ents]})

    def _planes_pending(self):
        """Does a filter want planes that no set holds yet?  (host-side check, two attribute reads per filter)"""
        have = {id(w) for st in self._plane_sets for w in st['filters']}
        return any(e[1] % 8 == 0 and e[4] % 8 == 0 and id(e[0]) not in have and (HF.planes_mode == 'all' or getattr(e[0], '_dsrl_want_planes', False))
                   for e in self._split_entries)

    def refresh_transposed_filters(self):
        if self._wt_table is not None and os.environ.get('DSRL_BATCHED_TRANSPOSE', '1') != '0':
            presplit = HF.f16_mode() and os.environ.get('DSRL_PRESPLIT', '1') != '0'
            if presplit and self._split_table is None:
                self._build_split_filters()
            # with pre-split filters nothing reads the fp32 transposes: the first launch then only measures (amax records), the second writes both split forms
            if presplit and self._amax_key is not None and self._amax_key == self._params_key() and self._fold_amax():
                pass            # round 5: the optimiser pass of the previous step left max |w| of every filter it wrote (dsrl_sgd_step_dev_segments)
            elif presplit and os.environ.get('DSRL_FILTER_AMAX_STREAM', '1') != '0':
                self.w_amax.zero_()
                HF.call('dsrl_conv2d_filters_amax_batched', self._amax_seg_table.data_ptr(), self._amax_segs, HF._stream())
            else:
                self.w_amax.zero_()
                HF.call('dsrl_conv2d_transpose_filters_batched', (self._amax_only_table if presplit else self._wt_table).data_ptr(), self._wt_rows, self._wt_tiles, HF._stream())
            if presplit:
                HF.call('dsrl_conv2d_split_filters_batched', self._split_table.data_ptr(), self._wt_rows, self._wt_tiles, HF._stream())
            planes = presplit and HF.planes_mode != 'off' and HF.get_conv_precision() in ('f16x3', 'f16x1')      # f16x1 reads the first plane only
            if planes:
                if HF.graph_keepalive is None and self._planes_pending():
                    self._build_plane_filters()         # first use, or more filters asked for planes since the last set was built (never inside a capture)
                for st in self._plane_sets:
                    HF.call('dsrl_conv2d_filter_planes_batched', st['table'].data_ptr(), st['rows'], st['tiles'], HF._stream())
            self.wt_valid, self.wt_fp32_valid, self.split_valid, self.planes_valid = True, not presplit, presplit, planes

    def zero_grad(self):
        self._fill_skipped = self.lazy_zero and self._all_claimed_last
        if not self._fill_skipped:
            self.g_flat.zero_()
        self._claimed.clear()
        self._pending = [c[2] for c in self.chunks]
        self._works = []
        self._reduced = set()
        if self.device.type == 'cuda':
            # The convs of this step defer their weight gradients to grouped launches at the end of backward (functional.WgradQueue) - unless the
            # chunk all-reduces are launched from gradient-ready notifications DURING backward (world > 1, eager): a deferred weight gradient
            # would report ready only at the flush and every collective would start behind backward; that path keeps the per-layer
            # side-stream launches, which also free x / dy layer by layer.
            if self.world == 1 or self.defer_collectives:
                HF.open_wgrad_queue()
            else:
                HF.wgrad_queue = None
            if HF.f16_mode():
                HF.amax_begin_step(self.device)     # operand-magnitude slots of this step (functional.amax_slot)

    def settle_grads(self):
        """After the backward pass (every gradient kernel enqueued or recorded): was it right not to zero the arena?  Remembers whether the next
        zero_grad() may skip the fill."""
        complete = len(self._claimed) == len(self.params)
        if self._fill_skipped and not complete:
            missing = [i for i in range(len(self.params)) if i not in self._claimed]
            raise HF.DsrlHipError(f'{len(missing)} parameter gradients were not written by their kernels in this pass (first: parameter #{missing[0]}) although the '
                                  'previous pass wrote all of them, and the gradient arena was not zeroed: the graph of the step changed - set DSRL_LAZY_ZERO_GRAD=0')
        self._all_claimed_last = complete
        self._fill_skipped = False

    def finish_reduction(self):
        """Waits (stream-wise) for the chunk all-reduces launched during backward; chunks whose hooks did not all fire
        (parameters unused in this step) are reduced here so that every rank issues the same collectives."""
        if self.device.type == 'cuda':
            HF.flush_wgrad_queue()          # the deferred weight gradients of this backward pass, as grouped grids
            HF.join_side_streams()          # weight gradients written on the side stream (functional.overlap_wgrad)
        if self.world == 1:
            return
        for ci, left in enumerate(self._pending):
            if left > 0:
                a, b, _ = self.chunks[ci]
                self._works.append(dist.all_reduce(self.g_flat[a:b], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
                self._pending[ci] = 0
        for w in self._works:
            w.wait()
        self._works = []

    # ------------------------------------------------------------------ two-phase reduction (split hipGraph capture, world > 1)
    def _ranges(self, idx):
        """chunk indices -> merged (start, end) element ranges: neighbouring chunks travel as one collective (larger messages)"""
        out = []
        for ci in sorted(idx):
            a, b, _ = self.chunks[ci]
            if out and out[-1][1] == a:
                out[-1][1] = b
            else:
                out.append([a, b])
        return out

    def ready_chunks(self):
        """Chunks whose every gradient has been produced so far in this backward pass and that have not been reduced yet."""
        return [ci for ci, left in enumerate(self._pending) if left == 0 and ci not in self._reduced]

    def reduce_chunks(self, idx):
        """Launches (asynchronously, behind what the current stream holds) the all-reduce of the given chunks; reduce_rest() waits."""
        if self.world == 1:
            return
        for a, b in self._ranges(idx):
            self._works.append(dist.all_reduce(self.g_flat[a:b], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        self._reduced.update(idx)

    def reduce_rest(self):
        """All chunks not reduced yet, then the current stream waits for every collective of the step."""
        if self.world == 1:
            return
        self.reduce_chunks([ci for ci in range(len(self.chunks)) if ci not in self._reduced])
        for w in self._works:
            w.wait()
        self._works = []
        self._pending = [0] * len(self.chunks)

    def reduce_all(self):
        """All chunk all-reduces of the gradient arena at once, behind whatever the current stream holds (the replay of a captured
        forward + backward); the caller's stream waits for them.  Used when backward ran from a hipGraph, where no hook fires."""
        if self.world == 1:
            return
        # nothing overlaps with these collectives, so they are as large as possible (ring all-reduce bandwidth over xGMI grows with the message
        # size): the whole 239.5 MB arena in pieces of DSRL_REDUCE_ALL_MB (default 256, i.e. one call)
        per = max(1, int(os.environ.get('DSRL_REDUCE_ALL_MB', '256'))) << 18          # floats
        works = [dist.all_reduce(self.g_flat[a:min(a + per, self.numel)], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                 for a in range(0, self.numel, per)]
        for w in works:
            w.wait()

    def reduce_chunked_and_step(self, hyper=None, hp=None, nchunks=None):
        """The default exchange behind a replayed (or deferred eager) backward pass with more than one rank: the gradient arena is cut into
        `nchunks` contiguous ranges (DSRL_REDUCE_CHUNKS, default 4), every range's all-reduce is launched at once on the collective stream, and the
        SGD kernel of range k runs on the compute stream as soon as ITS all-reduce is done - under the all-reduces of ranges k+1 .. (round 5; the
        single 239.5 MB call of round 4 exposed the whole optimiser pass behind the exchange).  Only element-wise kernels touch the ranges, so they
        need no parameter alignment beyond 16 bytes; no barrier kernel is co-resident with RCCL here - those are all inside the graph that has
        finished.  Bit-identical to reduce_all() + one SGD launch (same element-wise arithmetic).  Returns the chunk ranges (tests)."""
        n = max(1, int(os.environ.get('DSRL_REDUCE_CHUNKS', '4')) if nchunks is None else int(nchunks))
        per = _align(-(-self.numel // n), 1024)
        ranges = [(a, min(a + per, self.numel)) for a in range(0, self.numel, per)]
        works = []
        if self.world > 1:
            works = [dist.all_reduce(self.g_flat[a:b], op=dist.ReduceOp.SUM, group=self.pg, async_op=True) for a, b in ranges]
        fold = hyper is not None and self._fold_amax()
        if fold:
            self.w_amax.zero_()
        for i, (a, b) in enumerate(ranges):
            if works:
                works[i].wait()             # stream-ordered for RCCL: the compute stream waits for this range only
            self._sgd_range(a, b, hyper, hp)
        self._amax_key = self._params_key() if fold else None
        self._pending = [0] * len(self.chunks)
        if self.device.type == 'cuda':
            HF.amax_end_step(self.device)
        self.wt_valid = self.wt_fp32_valid = self.split_valid = self.planes_valid = False
        return ranges

    def sync_buffers(self):
        if self.world > 1 and self.broadcast_buffers_enabled:
            dist.broadcast(self.b_flat, 0, group=self.pg)

    # ------------------------------------------------------------------ optimiser
    def sgd_step(self, lr, momentum, weight_decay, hyper=None, reduce=True):
        """torch.optim.SGD(momentum, weight_decay).step() on the whole arena; gradients are averaged over ranks here.  `hyper`: a
        4-float device tensor (lr, momentum, weight_decay, 1/world) the kernel reads when it runs (graph-replayable) instead of the
        three host values."""
        if reduce:
            self.finish_reduction()
        fold = hyper is not None and self._fold_amax()
        if fold:
            self.w_amax.zero_()
        self._sgd_range(0, self.numel, hyper, (lr, momentum, weight_decay))
        self._amax_key = self._params_key() if fold else None
        if self.device.type == 'cuda':
            HF.amax_end_step(self.device)       # records asked for between steps (validation) come from the loose arena, not from the step's
        self.wt_valid = self.wt_fp32_valid = self.split_valid = self.planes_valid = False       # the filters changed: transposed / split copies and amax records are stale until the next refresh

    def _trainable_in_model_order(self):
        return [p for p in self.model.parameters() if p.requires_grad]

    def state_dict(self, lr=0.0, momentum=0.0, weight_decay=0.0, initial_lr=None):
        """The optimiser state in torch.optim.SGD's state_dict layout (what the reference saves as `optimizer_state_dict`,
        train_or_resume.py:276, and loads back at main.py:51-52): one param group over model.parameters() order, per-parameter
        `momentum_buffer` tensors (copies of the arena views, channels-last parameters returned in their logical NCHW shape)."""
        ps = self._trainable_in_model_order()
        state = {}
        for i, p in enumerate(ps):
            o = self.offsets[self._index[id(p)]]
            state[i] = {'momentum_buffer': self.m_flat.as_strided(p.shape, p.stride(), o).detach().clone().contiguous().cpu()}
        group = {'lr': lr, 'momentum': momentum, 'dampening': 0, 'weight_decay': weight_decay, 'nesterov': False, 'params': list(range(len(ps)))}
        if initial_lr is not None:
            group['initial_lr'] = initial_lr            # what a torch LR scheduler leaves in the group (PolynomialLR in the reference)
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, sd):
        """Accepts torch.optim.SGD's layout (a reference checkpoint or our own) and the round-1 arena dump."""
        if 'momentum_arena' in sd:
            if sd['numel'] != self.numel:
                raise ValueError('optimizer arena size mismatch')
            self.m_flat.copy_(sd['momentum_arena'])
            return
        ps = self._trainable_in_model_order()
        ids = [i for g in sd['param_groups'] for i in g['params']]
        if len(ids) != len(ps):
            raise ValueError(f'optimizer state has {len(ids)} parameters, the model {len(ps)}')
        self.m_flat.zero_()
        with torch.no_grad():
            for p, i in zip(ps, ids):
                st = sd['state'].get(i)
                buf = None if st is None else st.get('momentum_buffer')
                if buf is None:
                    continue                          # torch creates the buffer lazily: a parameter that never stepped has none
                if tuple(buf.shape) != tuple(p.shape):
                    raise ValueError(f'momentum buffer {tuple(buf.shape)} does not match parameter {tuple(p.shape)}')
                o = self.offsets[self._index[id(p)]]
                self.m_flat.as_strided(p.shape, p.stride(), o).copy_(buf.to(self.device))
