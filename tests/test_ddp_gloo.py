"""World-size-2 gloo test of the flat-arena gradient reducer (ddp.FlatParams) on CPU: the arena views alias the parameters,
rank 0's weights win at construction, chunked asynchronous all-reduce from the autograd hooks produces the mean gradient,
and the fused SGD kernel refuses to run without a GPU (no CPU fallback)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from dualsuperreslearningforsemseg_amd.ddp import FlatParams
        torch.manual_seed(100 + rank)                       # deliberately different init per rank
        model = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.ReLU(), torch.nn.Conv2d(8, 4, 1))
        model = model.to(memory_format=torch.channels_last)
        flat = FlatParams(model, chunk_bytes=64)           # tiny chunks -> several collectives
        assert len(flat.chunks) >= 2
        w0 = [p.detach().clone() for p in model.parameters()]
        gathered = [torch.zeros_like(flat.p_flat) for _ in range(world)]
        dist.all_gather(gathered, flat.p_flat)
        assert torch.equal(gathered[0], gathered[1]), 'constructor broadcast did not equalise the weights'
        for p in model.parameters():
            assert p.data_ptr() >= flat.p_flat.data_ptr() and p.grad.data_ptr() >= flat.g_flat.data_ptr()
        # per-rank batch; reference = single process on the concatenated batch with BN in eval (BN is unsynced in the reference)
        model.eval()
        torch.manual_seed(7)
        xs = torch.randn(world, 2, 3, 8, 8)
        flat.zero_grad()
        model(xs[rank]).square().mean().backward()
        flat.finish_reduction()
        mean_grad = flat.g_flat / world
        ref = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.ReLU(), torch.nn.Conv2d(8, 4, 1)).eval()
        ref.load_state_dict(model.state_dict())
        ref(xs.reshape(world * 2, 3, 8, 8)).square().mean().backward()
        for (n, p), (_, r) in zip(model.named_parameters(), ref.named_parameters()):
            o = flat.offsets[[id(q_) for q_ in flat.params].index(id(p))]
            got = mean_grad.as_strided(p.shape, p.stride(), o)
            assert torch.allclose(got, r.grad, rtol=1e-4, atol=1e-6), n
        # deferred form (a step replayed from a hipGraph): no collective from the hooks, reduce_all() afterwards gives the same sum
        hooked = flat.g_flat.clone()
        flat.defer_collectives = True
        flat.zero_grad()
        model(xs[rank]).square().mean().backward()
        assert len(flat._works) == 0, 'a hook launched a collective although they are deferred'
        flat.reduce_all()
        flat.defer_collectives = False
        assert torch.allclose(flat.g_flat, hooked, rtol=1e-6, atol=1e-7)
        # two-phase form (TrainStep's split capture): backward down to a detached cut, the chunks complete by then are all-reduced while the
        # rest of backward runs, the others behind it - same sums; and the chunk bookkeeping ends at exactly zero outstanding notifications
        flat.defer_collectives = True
        flat.zero_grad()
        h = model[1](model[0](xs[rank]))
        hd = h.detach().requires_grad_(True)
        model[3](model[2](hd)).square().mean().backward()
        ready = flat.ready_chunks()
        assert ready and len(ready) < len(flat.chunks), (ready, flat._pending)      # the last conv's chunks, not the first conv's
        flat.reduce_chunks(ready)
        torch.autograd.backward([h], [hd.grad])
        assert flat._pending == [0] * len(flat.chunks), flat._pending
        flat.reduce_rest()
        flat.defer_collectives = False
        assert torch.allclose(flat.g_flat, hooked, rtol=1e-6, atol=1e-7)
        # a parameter whose kernel writes its gradient straight into the arena (claim) is reported by written() only: torch runs its
        # post-accumulate hook as well, which must not count (it did until round 3: a chunk looked complete after half of its notifications)
        flat.zero_grad()
        p0 = flat.params[0]
        assert flat.claim(p0) and not flat.claim(p0)
        before = list(flat._pending)
        flat._make_hook(0)(p0)
        assert flat._pending == before
        flat.defer_collectives = True
        flat.written(p0)
        flat.defer_collectives = False
        assert flat._pending[flat._chunk_of[0]] == before[flat._chunk_of[0]] - 1
        # buffers travel from rank 0
        model[1].running_mean.fill_(float(rank + 1))
        flat.sync_buffers()
        assert float(model[1].running_mean[0]) == 1.0
        try:
            flat.sgd_step(0.1, 0.9, 0.0)
            q.put((rank, 'sgd ran on CPU'))
        except RuntimeError as e:
            q.put((rank, 'ok' if 'no CPU fallback' in str(e) else str(e)))
    except Exception as e:      # noqa: BLE001
        import traceback; q.put((rank, traceback.format_exc()[-600:]))
    finally:
        dist.destroy_process_group()


def _sgd_reference(p, g, buf, lr, momentum, weight_decay, grad_scale=1.0):
    """torch.optim.SGD's update on flat arenas: the CPU stand-in of dsrl_sgd_step (which has no CPU path) for the schedule test below"""
    with torch.no_grad():
        d = g * grad_scale + weight_decay * p
        buf.mul_(momentum).add_(d)
        p.sub_(lr * buf)


def _worker_trainstep(rank, world, port, q):
    """TrainStep's exchange-and-update section (`_finish`, the one-graph schedule of more than one rank) with two real gloo ranks: the gradient
    arena travels in DSRL_REDUCE_CHUNKS ranges, each range's SGD kernel follows ITS all-reduce.  The device kernel is replaced by the torch formula
    (the schedule, the ranges and the collectives are the product's); both ranks must end with identical parameters, equal to one torch.optim.SGD step on
    the mean gradient, for every chunk count including one that does not divide the arena."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from dualsuperreslearningforsemseg_amd import ddp, functional as HF
        from dualsuperreslearningforsemseg_amd.command_handlers.train_or_resume import TrainStep
        HF.sgd_step_ = _sgd_reference
        calls = {'n': 0}
        real_ar = ddp.dist.all_reduce

        def counted(*a, **k):
            calls['n'] += 1
            return real_ar(*a, **k)
        ddp.dist.all_reduce = counted
        for nch in (1, 3, 4):
            os.environ['DSRL_REDUCE_CHUNKS'] = str(nch)
            torch.manual_seed(5)
            mk = lambda: torch.nn.Sequential(torch.nn.Conv2d(3, 16, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(16, 16, 3, padding=1), torch.nn.ReLU(),      # noqa: E731
                                             torch.nn.Conv2d(16, 5, 1))
            model, ref = mk(), mk()
            ref.load_state_dict(model.state_dict())
            flat = ddp.FlatParams(model, chunk_bytes=256)
            step = TrainStep(model, flat, 1, 0.1, 1.0, 255, graph=False)
            flat.defer_collectives = True               # what TrainStep selects for a replayed step with more than one rank
            assert not step.split
            opt = torch.optim.SGD(ref.parameters(), lr=0.05, momentum=0.9, weight_decay=5e-4)
            torch.manual_seed(11)
            xs = torch.randn(3, world, 2, 3, 8, 8)
            for it in range(3):
                flat.zero_grad()
                model(xs[it, rank]).square().mean().backward()
                assert len(flat._works) == 0
                calls['n'] = 0
                step._finish((0.05, 0.9, 5e-4), False)
                assert calls['n'] == len(range(0, flat.numel, ddp._align(-(-flat.numel // nch), 1024))), (calls, nch)
                opt.zero_grad()
                ref(xs[it].reshape(world * 2, 3, 8, 8)).square().mean().backward()         # mean over both ranks' samples = mean of the per-rank means
                opt.step()
            gathered = [torch.zeros_like(flat.p_flat) for _ in range(world)]
            dist.all_gather(gathered, flat.p_flat)
            assert torch.equal(gathered[0], gathered[1]), 'ranks diverged'
            for (n, p), (_, r) in zip(model.named_parameters(), ref.named_parameters()):
                assert torch.allclose(p, r, rtol=2e-5, atol=1e-6), (nch, n, float((p - r).abs().max()))
        q.put((rank, 'ok'))
    except Exception:      # noqa: BLE001
        import traceback; q.put((rank, traceback.format_exc()[-900:]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_trainstep_chunked_exchange_and_update_world2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_trainstep, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == {0: 'ok', 1: 'ok'}, res


@pytest.mark.timeout(300)
def test_flat_arena_allreduce_world2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == {0: 'ok', 1: 'ok'}, res
