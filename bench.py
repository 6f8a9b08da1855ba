#!/usr/bin/env python3
"""Headline benchmark: stage-3 DSRL training images/sec at 256x512 -> 512x1024 (BASELINE.json), synthetic device-resident
Cityscapes-shaped batches, per-rank batch 8 (train_stage3_cmdline.json), one process per GPU, gradients all-reduced over
RCCL.  A "step" = forward (ResNet-101 + ASPP + SSSR/SISR decoders + feature transformers) + CE/MSE/FA losses + backward +
gradient reduction + SGD update + the per-iteration loss/NaN readback, exactly what train_or_resume() runs per batch.

    python bench.py [--gpus N --steps K --warmup W]           (N > 1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     - the dominant kernel (the MFMA implicit-GEMM conv that takes the most device time): achieved in-bounds (algorithmic)
                 TFLOP/s from HIP events recorded by the library around every launch, against the dense MFMA peak of the
                 arithmetic that kernel runs in: 157.3 TFLOP/s for fp32 MFMA, 2516.6/3 for bf16x3 and 2516.6/6 for bf16x6
                 (three / six bf16 MFMAs per algorithmic product - DESIGN.md section 4); every conv kernel family is listed;
  cpu_baseline - the numpy oracle (oracle/, kind "port") timed on this box's host cores on a bounded sample.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 matrix peak 157.3 TFLOP/s, dense bf16 2516.6 TFLOP/s.  A split-precision conv
# issues 3 (bf16x3) or 6 (bf16x6) bf16 MFMAs per algorithmic product, so its MFMA roof for ALGORITHMIC flops is the bf16 peak / 3 or / 6.
BF16_MFMA_PEAK_TFLOPS = 2516.6
MFMA_PEAK_TFLOPS = {0: 157.3, 1: BF16_MFMA_PEAK_TFLOPS / 3, 2: BF16_MFMA_PEAK_TFLOPS / 6}      # by arithmetic: fp32, bf16x3, bf16x6
ARITH_NAME = {0: 'fp32', 1: 'bf16x3', 2: 'bf16x6'}


def cpu_baseline(state_dict, batch=2, height=256, width=512):
    """The same workload on the host cores: numpy oracle (oracle/, a port of the reference arithmetic), whole stage-3 step =
    ResNet-101 + head forward, CE/MSE/FA losses and the full backward pass, fp32, B=2 at 256x512 (a bounded sample: ~20 s)."""
    import numpy as np
    import oracle as O
    rs = np.random.RandomState(1234)
    sd = {k: v.detach().float().cpu().numpy() for k, v in state_dict.items() if 'num_batches' not in k}
    x = rs.standard_normal((batch, 3, height, width)).astype(np.float32)
    org = rs.standard_normal((batch, 3, 2 * height, 2 * width)).astype(np.float32)
    tg = rs.randint(0, 19, (batch, 2 * height, 2 * width)).astype(np.uint8); tg[rs.uniform(size=tg.shape) < 0.1] = 255
    t0 = time.time()
    out = O.model_forward(sd, x, 3, True)
    O.total_loss(out, tg, org, 3)
    dt = time.time() - t0
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else os.cpu_count()
    try:                                        # threads the BLAS behind numpy actually uses
        from threadpoolctl import threadpool_info
        blas = [i.get('num_threads') for i in threadpool_info() if i.get('user_api') == 'blas']
        cores = max(blas) if blas else cores
    except Exception:
        pass
    return {'value': round(batch / dt, 4), 'unit': 'images/s', 'cores': cores, 'kind': 'port',
            'sample': f'numpy oracle (fp32, multi-threaded BLAS), whole stage-3 step without the optimizer update: ResNet-101 + head forward, '
                      f'CE/MSE/FA, full backward; B={batch} at {height}x{width}->{2 * height}x{2 * width}; one pass = {dt:.1f} s'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=8, help='per-rank batch (train_stage3_cmdline.json: 8)')
    ap.add_argument('--height', type=int, default=256)
    ap.add_argument('--width', type=int, default=512)
    ap.add_argument('--stage', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-prof', action='store_true', help='do not record per-launch HIP events')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if os.environ.get('DSRL_ALL_RANKS_ON_GPU0'):                       # rehearsal of the multi-rank path on a one-GPU box
        local = 0
        os.environ['DSRL_BN_FUSED'] = '0'     # several processes on one GPU: the fused BN kernels' device-wide barrier needs the GPU to itself
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f'--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...`')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('DSRL_DIST_BACKEND', 'nccl')          # 'nccl' is RCCL on ROCm; 'gloo' only to rehearse N ranks on one GPU
        if backend == 'nccl':
            dist.init_process_group('nccl', init_method='env://', world_size=world, rank=rank, device_id=dev)
        else:
            dist.init_process_group(backend, init_method='env://', world_size=world, rank=rank)

    import dualsuperreslearningforsemseg_amd as D
    from dualsuperreslearningforsemseg_amd import _lib, settings
    from dualsuperreslearningforsemseg_amd.command_handlers.train_or_resume import SyntheticCityscapes, TrainStep
    from dualsuperreslearningforsemseg_amd.datasets.Cityscapes import settings as cs
    from dualsuperreslearningforsemseg_amd.ddp import FlatParams
    lib = _lib.load()

    torch.manual_seed(settings.RANDOM_SEED)                      # identical initial weights on every rank (train_or_resume.py:31)
    model = D.DSRL(args.stage, cs)
    with torch.no_grad():                                        # random-init stand-in for the ImageNet backbone: un-zero the residual BNs
        for m in model.modules():
            if hasattr(m, 'bn3'):
                m.bn3.weight.fill_(0.5)
    model = model.to(dev).to(memory_format=torch.channels_last).train()
    flat = FlatParams(model)
    step = TrainStep(model, flat, args.stage, 0.1, 1.0, cs.IGNORE_CLASS_LABEL)
    data = SyntheticCityscapes(args.batch, (args.height, args.width), dev, rank=rank, length=1)
    (img, org), (tgt, _) = next(iter(data))
    hp = dict(lr=0.006, momentum=0.9, weight_decay=5e-4)        # train_stage3_cmdline.json

    host_ms = []

    def run(n):
        # train_or_resume()'s loop: iteration k's loss/NaN readback is collected after iteration k+1 has been enqueued
        last = None
        for _ in range(n):
            step.enqueue(img, org, tgt, hp['lr'], hp['momentum'], hp['weight_decay'], True)
            host_ms.append(step.host_enqueue_s * 1e3)
            while step.pending() > 1:
                last = step.collect()
        while step.pending():
            last = step.collect()
        return last

    run(args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    PROF_STRIDE = 7         # the timed region brackets every 7th conv launch with HIP events (an event pair costs ~3 us of device time;
                            # 7 is co-prime with the conv launches per step, so all layers are sampled evenly over the steps)
    if not args.no_prof:
        lib.dsrl_prof_enable(PROF_STRIDE)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = run(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt)
        # data-parallel invariant: after K identical-seed steps every rank must hold bit-identical parameters
        chk = torch.stack([flat.p_flat.double().sum(), flat.p_flat.double().abs().sum()])
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert torch.equal(lo, hi), f'ranks diverged: {lo.tolist()} vs {hi.tolist()}'

    def read_prof(nsteps, stride=1):
        fams = []
        for fam in range(9):                 # family = 3 * arithmetic + pass (include/dsrl_hip.h)
            n = ctypes.c_int64(0); ms = ctypes.c_double(0); fl = ctypes.c_double(0)
            _lib.check(lib.dsrl_prof_read(fam, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl)), 'dsrl_prof_read')
            by = ctypes.c_double(0)
            _lib.check(lib.dsrl_prof_read_bytes(fam, ctypes.byref(by)), 'dsrl_prof_read_bytes')
            if n.value:
                fams.append((lib.dsrl_prof_kernel_name(fam).decode(), n.value, ms.value, fl.value, fam // 3, by.value))
        lib.dsrl_prof_enable(0)
        name, n, ms, fl, arith, by = max(fams, key=lambda f: f[2])          # dominant = most device time
        ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        peak = MFMA_PEAK_TFLOPS[arith]
        return {'achieved': round(ach, 2), 'peak': round(peak, 1), 'frac': round(ach / peak, 4), 'kernel': name, 'arithmetic': ARITH_NAME[arith],
                'launches_per_step': n * stride // max(nsteps, 1), 'launches_timed': n,
                'avg_launch_ms': round(ms / max(n, 1), 5), 'avg_launch_gflop': round(fl / max(n, 1) / 1e9, 3),
                'algorithmic_bytes_per_launch': int(by / max(n, 1)), 'kernel_ms_per_step': round(ms * stride / nsteps, 3),
                'all_mfma_kernels': {f[0]: {'ms_per_step': round(f[2] * stride / nsteps, 3), 'tflops': round(f[3] / (f[2] * 1e-3) / 1e12, 2) if f[2] > 0 else 0.0,
                                            'peak': round(MFMA_PEAK_TFLOPS[f[4]], 1), 'algorithmic_bytes_per_launch': int(f[5] / max(f[1], 1)),
                                            'frac': round(f[3] / (f[2] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS[f[4]], 4) if f[2] > 0 else 0.0}
                                     for f in fams}}

    from dualsuperreslearningforsemseg_amd import functional as HF
    conv_arith = {'fp32': 'fp32 MFMA', 'bf16x3': 'bf16x3 split, fp32 accumulate', 'bf16x6': 'bf16x6 split (fp32-equivalent), fp32 accumulate',
                  'mixed': 'forward bf16x6 (fp32-equivalent), dgrad/wgrad bf16x3; fp32 storage and accumulation'}[HF.get_conv_precision()]
    roof = None
    if not args.no_prof:
        timed = read_prof(args.steps, PROF_STRIDE)
        roof = {'bound': 'mfma', 'peak': None, 'unit': 'TFLOP/s', 'traffic': None}
        if HF.overlap_wgrad:
            # In the timed region weight-gradient kernels run on a side stream concurrently with data-gradient / BN kernels, so a
            # kernel's event-to-event time includes the share of the GPU it gave away. The per-kernel roofline is therefore taken
            # from a few extra steps with that overlap disabled (kernels run one at a time); both figures are reported.
            HF.overlap_wgrad = False
            run(1); torch.cuda.synchronize()
            lib.dsrl_prof_enable(1)
            excl_steps = 5
            run(excl_steps); torch.cuda.synchronize()
            excl = read_prof(excl_steps)
            HF.overlap_wgrad = True
            roof.update(excl)
            roof['measured'] = f'{excl_steps} extra steps right after the timed region, weight-gradient side stream disabled (exclusive kernel execution)'
            roof['timed_region_with_stream_overlap'] = {k: timed[k] for k in ('kernel', 'achieved', 'peak', 'frac', 'avg_launch_ms', 'launches_timed', 'kernel_ms_per_step', 'all_mfma_kernels')}
            roof['timed_region_with_stream_overlap']['sampling'] = f'every {PROF_STRIDE}th conv launch of the timed region'
        else:
            roof.update(timed)
            roof['measured'] = 'timed region'

    if roof is not None:
        try:        # HBM-side bytes per launch from the committed rocprofv3 --pmc passes of this same command (not collectable in-process)
            pmc = json.load(open(os.path.join(ROOT, 'profiles', 'round1_pmc_traffic.json')))
            if (args.stage, args.height, args.width, args.batch) == (3, 256, 512, 8) and roof.get('kernel') in pmc:
                roof['traffic'] = pmc[roof['kernel']]['hbm_bytes_per_launch']
                roof['traffic_source'] = 'profiles/round1_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH_SIZE x2 on gfx950)'
        except Exception:
            pass

    # the same step under the stricter conv arithmetics, a few steps each (single GPU only): fp32-equivalent bf16x6 everywhere and
    # exact-product fp32 MFMA.  The headline `value` is the default 'mixed' mode (bf16x6 forward, bf16x3 backward), see DESIGN.md 3b.
    by_arith = None
    if world == 1 and not args.no_prof and HF.get_conv_precision() == 'mixed':
        by_arith = {}
        for mode in ('bf16x6', 'fp32'):
            HF.set_conv_precision(mode)
            run(3); torch.cuda.synchronize()
            t1 = time.perf_counter(); run(10); torch.cuda.synchronize()
            by_arith[mode] = round(args.batch * 10 / (time.perf_counter() - t1), 1)
        HF.set_conv_precision(None)

    if rank == 0:
        gb = args.batch * world
        line = {
            'metric': 'stage-3 train images/sec at 256x512->512x1024' if (args.stage, args.height, args.width) == (3, 256, 512)
                      else f'stage-{args.stage} train images/sec at {args.height}x{args.width}',
            'value': round(gb * args.steps / elapsed, 3), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * elapsed / args.steps, 3), 'host_enqueue_ms_per_step': round(sorted(host_ms[args.warmup:args.warmup + args.steps])[args.steps // 2], 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'conv_arithmetic': conv_arith, 'images_per_s_by_conv_arithmetic': by_arith, 'data': 'synthetic',
            'config': {'workload': f'DSRL stage {args.stage} (ResNet-101 OS16 + ASPP + SSSR/SISR decoders + FA loss), full train step, '
                                   f'random-init weights, {args.height}x{args.width} input -> {2 * args.height}x{2 * args.width} logits',
                       'global_batch': gb, 'per_gpu_batch': args.batch, 'parallelism': f'dp{world}', 'optimizer': 'SGD m0.9 wd5e-4 lr0.006',
                       'losses_last_step': [round(v, 5) for v in losses]},
            'roofline': roof,
        }
        if not args.no_cpu_baseline and world == 1:
            line['cpu_baseline'] = cpu_baseline(model.state_dict())
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
