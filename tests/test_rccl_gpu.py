"""The N > 1 training path against real RCCL kernels on ONE GPU (`-m gpu`): a 1-rank 'nccl' process group with ddp.FlatParams told
that the world size is 2, so everything the multi-GPU step does is active - constructor + per-step buffer broadcasts, the chunked
all-reduces (launched from gradient-ready notifications in the eager path, behind the hipGraph replay in the graph path), the
1/world factor in the SGD kernel, the fused-BN block budget.  A 1-rank all-reduce is an identity, so the loss trajectory must equal
a plain single-process run at HALF the learning rate (the SGD kernel divides the summed gradient by the claimed world size).
Reference behaviour: command_handlers/train_or_resume.py:27-44, 105-106 (DDP mean-reduced gradients, rank-0 buffer broadcast)."""
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
STEPS = 7


def _make(world_claim, graph, lr):
    import dualsuperreslearningforsemseg_amd as D
    from dualsuperreslearningforsemseg_amd import ddp, functional as HF
    from dualsuperreslearningforsemseg_amd.command_handlers.train_or_resume import SyntheticCityscapes, TrainStep
    from dualsuperreslearningforsemseg_amd.datasets.Cityscapes import settings as cs
    torch.manual_seed(54321)
    model = D.DSRL(3, cs)
    with torch.no_grad():
        for m in model.modules():
            if hasattr(m, 'bn3'):
                m.bn3.weight.fill_(0.5)
    model = model.to(DEV).to(memory_format=torch.channels_last).train()
    if world_claim > 1:
        real = ddp.dist.get_world_size
        ddp.dist.get_world_size = lambda group=None: world_claim       # the collectives still run on the 1-rank group
        try:
            flat = ddp.FlatParams(model, chunk_bytes=16 << 20)
        finally:
            ddp.dist.get_world_size = real
        assert flat.world == world_claim and len(flat._hooks) > 0 and len(flat.chunks) >= 8
    else:
        flat = ddp.FlatParams(model)
    HF.set_dropout_seed(4242)
    step = TrainStep(model, flat, 3, 0.1, 1.0, cs.IGNORE_CLASS_LABEL, graph=graph)
    HF.set_bn_fused_max_blocks(128)         # one summation order of the fused-BN partials in all three runs (the eager N > 1 path caps it at 128)
    (img, org), (tgt, _) = next(iter(SyntheticCityscapes(2, (64, 128), torch.device(DEV), length=1)))
    # weight decay 0: then "sum over a 1-rank group, times 1/2" at lr is EXACTLY lr/2 (powers of two), and the trajectories can be compared tightly
    hist = [step(img, org, tgt, lr, 0.9, 0.0, True)[0] for _ in range(STEPS)]
    torch.cuda.synchronize()
    return hist, flat, step


def test_rccl_reduction_paths_match_half_lr_single_process():
    import torch.distributed as dist
    from dualsuperreslearningforsemseg_amd import ddp, functional as HF
    # This test compares SCHEDULES (hook-launched chunks, one graph + chunked exchange, two graphs) through 7-step trajectories, which needs every run to
    # form the same sums in the same order: a random-init batch-2 net amplifies a last-bit difference to 1e-3 within two steps (round 5: measured).  bn3's
    # backward sums from the next block's accumulating dgrad (default since round 5) are one such difference - the block in front of the two-graph cut has no
    # "next block" inside its graph and takes the barrier kernel instead - so they are off here; the default setting is covered by the graph == eager tests.
    # The shared gradient buffer of layer1's three consumers (round 5) is another: the two-graph schedule cuts that tensor into a leaf and sums in another order.
    shared_was, HF.bn_bwd_stats_shared = HF.bn_bwd_stats_shared, False
    outer_was, HF.outer_grad_slot = HF.outer_grad_slot, False
    try:
        _reduction_paths_body(dist, ddp, HF)
    finally:
        HF.bn_bwd_stats_shared = shared_was
        HF.outer_grad_slot = outer_was


def _reduction_paths_body(dist, ddp, HF):
    # two single-process references at half the learning rate: with the grouped weight-gradient launches (what the graph paths run) and with
    # the per-layer side-stream launches (what the eager N > 1 path runs, so that its chunk all-reduces start during backward: ddp.FlatParams.zero_grad)
    plains = {}
    for grouped in (True, False):
        was = HF.group_wgrad
        HF.group_wgrad = grouped
        try:
            hist0, flat0, step0 = _make(1, False, 0.003)
        finally:
            HF.group_wgrad = was
        plains[grouped] = (hist0, flat0.p_flat.clone())
        step0.release()
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1, device_id=torch.device(DEV))
    calls = {'all_reduce': 0, 'broadcast': 0}
    real_ar, real_bc = ddp.dist.all_reduce, ddp.dist.broadcast

    def counted_ar(*a, **k):
        calls['all_reduce'] += 1
        return real_ar(*a, **k)

    def counted_bc(*a, **k):
        calls['broadcast'] += 1
        return real_bc(*a, **k)
    ddp.dist.all_reduce, ddp.dist.broadcast = counted_ar, counted_bc
    import os
    finals = {}
    try:
        for graph, split in ((False, '1'), (True, '1'), (True, '0')):
            os.environ['DSRL_GRAPH_SPLIT'] = split
            calls['all_reduce'] = calls['broadcast'] = 0
            hist, flat, step = _make(2, graph, 0.006)
            assert HF.bn_fused_barrier_timeouts() == 0
            # every step reduces the whole gradient arena exactly once - chunk by chunk from the gradient-ready notifications in the eager
            # path; in the graph path as TWO collectives: the chunks complete at the layer3 / layer4 cut between the two graphs of a step
            # (beside the second one), the rest behind it - and broadcasts the BN buffers once (+2 at construction)
            # (one-graph schedule, round 5: the arena travels as DSRL_REDUCE_CHUNKS = 4 ranges whose SGD kernels run under the later ranges' all-reduces)
            assert calls['all_reduce'] == STEPS * ((2 if split == '1' else 4) if graph else len(flat.chunks)), (calls, len(flat.chunks))
            finals[(graph, split)] = flat.p_flat.clone()
            if graph and split == '1':
                c = next(iter(step._graphs.values()))
                done = sum(flat.chunks[ci][1] - flat.chunks[ci][0] for ci in c.ready)
                assert step.split and c.graph_b is not None and 0.3 < done / flat.numel < 0.7, (c.ready, done / flat.numel)
            assert calls['broadcast'] == STEPS + 2, calls
            if graph:
                assert step.graph_replays == STEPS - step.GRAPH_WARMUP and flat.defer_collectives
            else:
                assert step.graph_replays == 0 and not flat.defer_collectives
            plain, p_plain = plains[graph]
            worst = max(abs(a - b) / max(abs(a), 1e-6) for u, v in zip(plain, hist) for a, b in zip(u, v))
            assert worst < 1e-5, (graph, worst, plain[-1], hist[-1])
            assert np.isfinite(hist[-1]).all()
            rel = float((flat.p_flat - p_plain).norm() / p_plain.norm())
            assert rel < 1e-6, (graph, rel)
            step.release()
        # the two-graph step (collectives between and behind the graphs) and the one-graph step (one collective behind it) are the same arithmetic
        assert torch.equal(finals[(True, '1')], finals[(True, '0')])
    finally:
        os.environ.pop('DSRL_GRAPH_SPLIT', None)
        ddp.dist.all_reduce, ddp.dist.broadcast = real_ar, real_bc
        HF.set_bn_fused_max_blocks(None)
        dist.destroy_process_group()


def test_graph_replay_matches_eager_bitwise():
    """A hipGraph-replayed training step (device-resident dropout key, device-resident LR, self-resetting BN barrier) reproduces the
    eager launches bit for bit, including the fresh dropout masks of every iteration and a learning-rate change without re-capture."""
    from dualsuperreslearningforsemseg_amd import functional as HF
    import dualsuperreslearningforsemseg_amd as D
    from dualsuperreslearningforsemseg_amd import ddp
    from dualsuperreslearningforsemseg_amd.command_handlers.train_or_resume import SyntheticCityscapes, TrainStep
    from dualsuperreslearningforsemseg_amd.datasets.Cityscapes import settings as cs
    res = {}
    for graph in (False, True):
        torch.manual_seed(54321)
        model = D.DSRL(3, cs)
        with torch.no_grad():
            for m in model.modules():
                if hasattr(m, 'bn3'):
                    m.bn3.weight.fill_(0.5)
        model = model.to(DEV).to(memory_format=torch.channels_last).train()
        flat = ddp.FlatParams(model)
        HF.set_dropout_seed(777)
        was = HF.overlap_wgrad
        HF.overlap_wgrad = False            # the capture is linear; compare with the same launch order
        try:
            step = TrainStep(model, flat, 3, 0.1, 1.0, cs.IGNORE_CLASS_LABEL, graph=graph)
            batches = list(SyntheticCityscapes(2, (64, 128), torch.device(DEV), length=6, distinct=2))
            hist = [step(img, org, tgt, 0.006 if i < 4 else 0.002, 0.9, 5e-4, True)[0] for i, ((img, org), (tgt, _)) in enumerate(batches)]
        finally:
            HF.overlap_wgrad = was
        torch.cuda.synchronize()
        res[graph] = (hist, flat.p_flat.clone(), flat.b_flat.clone(), model.state_dict()['feature_extractor.backbone.bn1.num_batches_tracked'].item())
        if graph:
            assert step.graph_replays == 6 - step.GRAPH_WARMUP
        step.release()
    assert res[False][0] == res[True][0], (res[False][0], res[True][0])
    assert torch.equal(res[False][1], res[True][1]) and torch.equal(res[False][2], res[True][2])
    assert res[False][3] == res[True][3] == 6
    # dropout really changes from step to step (same batch 0 at steps 0, 2, 4 but different losses already at step 2 vs a frozen key is
    # covered by the bitwise equality with the eager run, whose key advances on the host)


def test_graph_amax_arena_outlives_evaluation_passes():
    """What a captured step references must stay alive and in place.  (1) The captured step bakes in the addresses of the amax arena (its zero fill, the
    records the kernels max into and read).  Validation runs
    eagerly between epochs and asks for thousands of records: they must come from another arena, and the captured one must stay alive and in
    place (it used to be replaced after ~10 validation batches, and every later replay wrote into freed memory).  (2) The filter-plane
    table and arenas of the per-step filter pass: a second batch shape that wants planes for more filters adds a set, it does not rebuild the first.
    Graph run == eager run bit for bit across training steps, > 2048 records of evaluation forwards, steps at the second shape, and more steps at the first."""
    from dualsuperreslearningforsemseg_amd import functional as HF
    import dualsuperreslearningforsemseg_amd as D
    from dualsuperreslearningforsemseg_amd import ddp
    from dualsuperreslearningforsemseg_amd.command_handlers.train_or_resume import SyntheticCityscapes, TrainStep
    from dualsuperreslearningforsemseg_amd.datasets.Cityscapes import settings as cs
    HF.set_conv_precision(None)
    res = {}
    min_was, mode_was = HF.planes_min_elems, HF.planes_mode
    HF.planes_min_elems, HF.planes_mode = 200000, 'auto'          # 64x128: the decoder 3x3 convs want planes; 128x256: layer1 and the dilated ASPP convs too
    for graph in (False, True):
        torch.manual_seed(999)
        model = D.DSRL(3, cs).to(DEV).to(memory_format=torch.channels_last).train()
        flat = ddp.FlatParams(model)
        HF.set_dropout_seed(5)
        was = HF.overlap_wgrad
        HF.overlap_wgrad = False
        try:
            step = TrainStep(model, flat, 3, 0.1, 1.0, cs.IGNORE_CLASS_LABEL, graph=graph)
            batches = list(SyntheticCityscapes(2, (64, 128), torch.device(DEV), length=3, distinct=1))
            hist = [step(img, org, tgt, 0.006, 0.9, 5e-4, True)[0] for (img, org), (tgt, _) in batches]
            dev_ = flat.device
            arena = HF._amax_arena[dev_][0]
            ptr, pinned = arena.data_ptr(), HF._amax_arena[dev_][3]
            assert pinned or not graph          # (a capture of an earlier test of this process may have pinned it already)
            model.eval()
            (img, org), (tgt, _) = batches[0]
            asked = 0
            with torch.no_grad():
                while asked <= 2 * HF._AMAX_SLOTS:
                    before = HF._amax_loose.get(dev_, [None, 0])[1]
                    loose_before = HF._amax_loose.get(dev_, [None, 0])[0]
                    model(img)
                    lo = HF._amax_loose[dev_]
                    asked += (lo[1] - before) if lo[0] is loose_before else lo[1] + (HF._AMAX_SLOTS - before)
            model.train()
            assert HF._amax_arena[dev_][0] is arena and arena.data_ptr() == ptr and HF._amax_arena[dev_][1] < HF._AMAX_SLOTS
            # a larger batch shape in between wants plane operands for MORE filters (threshold lowered for this test): their planes come as an additional
            # set - the table and arenas the first graph was captured with stay where they are (ddp.FlatParams._build_plane_filters)
            sets_before = [(st['table'].data_ptr(), st['arenas'][0].data_ptr()) for st in flat._plane_sets]
            big = list(SyntheticCityscapes(2, (128, 256), torch.device(DEV), length=3, distinct=1))
            hist += [step(img_, org_, tgt_, 0.006, 0.9, 5e-4, True)[0] for (img_, org_), (tgt_, _) in big]
            assert [(st['table'].data_ptr(), st['arenas'][0].data_ptr()) for st in flat._plane_sets][:len(sets_before)] == sets_before
            assert len(flat._plane_sets) > len(sets_before) >= 1
            hist += [step(img, org, tgt, 0.006, 0.9, 5e-4, True)[0] for (img, org), (tgt, _) in batches]
        finally:
            HF.overlap_wgrad = was
        torch.cuda.synchronize()
        res[graph] = (hist, flat.p_flat.clone())
        step.release()
    HF.planes_min_elems, HF.planes_mode = min_was, mode_was
    assert all(np.isfinite(v) for h in res[True][0] for v in h)
    assert res[False][0] == res[True][0] and torch.equal(res[False][1], res[True][1])


def test_failed_capture_falls_back_to_eager(monkeypatch):
    """If the hipGraph capture fails, the TrainStep keeps training with eager launches (same results as a TrainStep that never tried)."""
    from dualsuperreslearningforsemseg_amd import functional as HF
    import dualsuperreslearningforsemseg_amd as D
    from dualsuperreslearningforsemseg_amd import ddp
    from dualsuperreslearningforsemseg_amd.command_handlers.train_or_resume import SyntheticCityscapes, TrainStep
    from dualsuperreslearningforsemseg_amd.datasets.Cityscapes import settings as cs
    res = {}
    for fail in (False, True):
        if fail:
            monkeypatch.setenv('DSRL_GRAPH_FAIL_TEST', '1')
        torch.manual_seed(54321)
        model = D.DSRL(3, cs)
        with torch.no_grad():
            for m in model.modules():
                if hasattr(m, 'bn3'):
                    m.bn3.weight.fill_(0.5)
        model = model.to(DEV).to(memory_format=torch.channels_last).train()
        flat = ddp.FlatParams(model)
        HF.set_dropout_seed(99)
        step = TrainStep(model, flat, 3, 0.1, 1.0, cs.IGNORE_CLASS_LABEL, graph=fail)
        (img, org), (tgt, _) = next(iter(SyntheticCityscapes(2, (64, 128), torch.device(DEV), length=1)))
        res[fail] = [step(img, org, tgt, 0.006, 0.9, 5e-4, True)[0] for _ in range(5)]
        if fail:
            assert step.graph_replays == 0 and not step.use_graph
        step.release()
    assert res[False] == res[True], (res[False][-1], res[True][-1])


@pytest.mark.parametrize('graph', [False, True])
def test_optimiser_pass_leaves_the_filter_magnitudes(graph, monkeypatch):
    """Round 5: dsrl_sgd_step_dev_segments updates the arena segment by segment and leaves max |w| of every conv filter it wrote in the filter's amax record,
    so the next step's filter pass starts with the split (DSRL_SGD_AMAX=0: the separate measuring sweep of rounds 3-4).  Same records bit for bit, hence the
    same split filters, losses and parameters; torch code that writes parameters between two replays (here: a scaled arena) invalidates the optimiser's
    magnitudes and the step measures again (ddp.FlatParams._params_key / ensure_filter_amax)."""
    from dualsuperreslearningforsemseg_amd import functional as HF
    res = {}
    for fold in ('1', '0'):
        monkeypatch.setenv('DSRL_SGD_AMAX', fold)
        hist, flat, step = _make(1, graph, 0.006)
        import dualsuperreslearningforsemseg_amd.command_handlers.train_or_resume as T
        from dualsuperreslearningforsemseg_amd.datasets.Cityscapes import settings as cs
        (img, org), (tgt, _) = next(iter(T.SyntheticCityscapes(2, (64, 128), torch.device(DEV), length=1)))
        assert (flat._amax_key is not None) == (fold == '1' and graph)      # eager launches take host hyper-parameters (dsrl_sgd_step): no segments, the step measures
        with torch.no_grad():
            flat.p_flat.mul_(1.5)               # an external write: every filter magnitude the optimiser left is stale now
        more = [step(img, org, tgt, 0.006, 0.9, 0.0, True)[0] for _ in range(3)]
        torch.cuda.synchronize()
        assert np.isfinite(more[-1]).all()
        flat.refresh_transposed_filters()       # fold: keeps the optimiser's records; else: measures the current weights - the same numbers
        torch.cuda.synchronize()
        # a record = 16 shards (which shard a block maxes into depends on its block index): the magnitude is the maximum over the shards
        res[fold] = (hist + more, flat.p_flat.clone(), flat.w_amax.view(-1, 16, 16)[:, :, 0].max(dim=1).values.clone(), flat.wsplit_flat.clone())
        step.release()
    assert res['1'][0] == res['0'][0], (res['1'][0][-1], res['0'][0][-1])
    for a, b in zip(res['1'][1:], res['0'][1:]):
        assert torch.equal(a, b)
