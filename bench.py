#!/usr/bin/env python3
"""Headline benchmark: stage-3 DSRL training images/sec at 256x512 -> 512x1024 (BASELINE.json), synthetic device-resident
Cityscapes-shaped batches, per-rank batch 8 (train_stage3_cmdline.json), one process per GPU, gradients all-reduced over
RCCL.  A "step" = forward (ResNet-101 + ASPP + SSSR/SISR decoders + feature transformers) + CE/MSE/FA losses + backward +
gradient reduction + SGD update + the per-iteration loss/NaN readback, exactly what train_or_resume() runs per batch.

    python bench.py [--gpus N --steps K --warmup W]

With --gpus N > 1 and no WORLD_SIZE in the environment the script starts its own N ranks (python -m torch.distributed.run, one
per GPU, rendezvous on 127.0.0.1) BEFORE anything touches the GPU and exits with their return code; launched by
torch.distributed.run it reads RANK / LOCAL_RANK / WORLD_SIZE from the environment.

Prints ONE JSON line on rank 0 (contract in the task statement).  `value` is measured with the DEFAULT conv arithmetic, the
fp32-equivalent "f16x3" split (two fp16 terms of the per-tensor scaled operands, three MFMAs per product; fp32 storage, fp32
accumulation, error vs fp64 equal to exact-product fp32 MFMA: DESIGN.md section 3b); the same step under the other arithmetics -
"bf16x6" (round 2's fp32-equivalent split), "mixed", exact-product "fp32" MFMA and the reduced-precision "f16x1" (one fp16 MFMA per
product: what apex O1 / O2 compute) - is timed over the same step counts and listed in `images_per_s_by_conv_arithmetic`.  Extra objects:
  roofline       - the dominant kernel (the MFMA implicit-GEMM conv family that takes the most device time): achieved in-bounds
                   (algorithmic) TFLOP/s from HIP events recorded by the library around every launch, against the dense MFMA peak
                   of the arithmetic that kernel runs in: 157.3 TFLOP/s for fp32 MFMA, 2516.6/3 for f16x3 / bf16x3 and 2516.6/6 for bf16x6
                   (three / six 16-bit MFMAs per algorithmic product - DESIGN.md section 3b), 2516.6 for f16x1; every conv kernel family is listed;
                   `decoder_stack`: the decoder-head conv stack (north_star's 30 % target) per pass, each conv timed on its own;
  roofline_hbm   - the memory-bound kernels of the step (CE, ConvTranspose, MSE, large BatchNorm, SGD): algorithmic bytes / time
                   against the 8 TB/s HBM peak;
  cpu_baseline   - the same training step on this box's host cores from stock torch.nn CPU modules (what the reference's
                   `--device cpu` path executes, oracle/torch_cpu_model.py), plus the numpy oracle as `cpu_baseline_numpy_port` and
                   `cpu_baseline_c1`: BASELINE.json configs[0] (stage 1, batch 2, 128x256) through the same stock-torch CPU graph.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 matrix peak 157.3 TFLOP/s, dense bf16 2516.6 TFLOP/s, HBM 8 TB/s.  A split-precision
# conv issues 3 (bf16x3) or 6 (bf16x6) bf16 MFMAs per algorithmic product, so its MFMA roof for ALGORITHMIC flops is the bf16 peak / 3 or / 6.
BF16_MFMA_PEAK_TFLOPS = 2516.6
MFMA_PEAK_TFLOPS = {0: 157.3, 1: BF16_MFMA_PEAK_TFLOPS / 3, 2: BF16_MFMA_PEAK_TFLOPS / 6, 3: BF16_MFMA_PEAK_TFLOPS / 3, 4: BF16_MFMA_PEAK_TFLOPS}      # by arithmetic: fp32, bf16x3, bf16x6, f16x3, f16x1
ARITH_NAME = {0: 'fp32', 1: 'bf16x3', 2: 'bf16x6', 3: 'f16x3', 4: 'f16x1'}
HBM_PEAK_GBS = 8000.0
ARITH_TEXT = {'fp32': 'exact-product fp32 MFMA', 'bf16x3': 'bf16x3 split, fp32 accumulate', 'bf16x6': 'bf16x6 split (fp32-equivalent), fp32 accumulate',
              'mixed': 'forward bf16x6 (fp32-equivalent), dgrad/wgrad bf16x3 (reduced-precision gradients); fp32 storage and accumulation',
              'f16x3': 'f16x3 split (two fp16 terms of the per-tensor power-of-two scaled operands, 3 MFMAs per product; fp32-equivalent: error vs fp64 '
                       'equal to exact-product fp32 MFMA), fp32 storage and accumulation',
              'f16x1': 'f16x1 (ONE fp16 term of the per-tensor scaled operands, one MFMA per product, fp32 storage and accumulation: the reduced-precision '
                       'arithmetic of apex O1 / O2)'}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=8, help='per-rank batch (train_stage3_cmdline.json: 8)')
    ap.add_argument('--height', type=int, default=256, help='input height (BASELINE config 5: --height 512 --width 1024)')
    ap.add_argument('--width', type=int, default=512)
    ap.add_argument('--stage', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-prof', action='store_true', help='skip the per-kernel roofline passes and the per-arithmetic runs')
    ap.add_argument('--no-config5', action='store_true', help="skip the bounded BASELINE config-5 run (512x1024 -> 1024x2048, 8 steps) of the default line")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` as the driver invokes it: become the launcher of N ranks.  Nothing here imports torch.cuda or
    touches the GPU, and nothing is exec'ed: the ranks are children of a plain subprocess."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '4')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


# ----------------------------------------------------------------------------------------------------------------- CPU baselines
def host_cores():
    """Cores this process may really use: the affinity mask, cut down to the cgroup CPU quota when there is one (a GPU box exposes all
    of the host's logical CPUs in the mask but grants a share of them: threads beyond the share only fight each other)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    if n > 32 and not os.environ.get('DSRL_CPU_BASELINE_THREADS'):
        n = 16          # no quota visible on a many-core host: the documented CPU share of a one-GPU box (both CPU legs are LIMITED to this count)
    return int(os.environ.get('DSRL_CPU_BASELINE_THREADS', n))


def cpu_baseline_torch(state_dict, stage, batch=2, height=256, width=512):
    """The same training step from stock torch.nn CPU modules (ATen / oneDNN): the arithmetic the reference's `--device cpu` path
    runs (utils.py:259-260).  One warm-up + one timed step of B=2 at 256x512 (bounded: ~10 s on 8 cores, less on a GPU box's host)."""
    from oracle.torch_cpu_model import time_train_step
    cores = host_cores()
    ips, threads, dt, n, med = time_train_step(state_dict, batch, height, width, stage, threads=cores, repeats=4, budget_s=12.0)
    return {'value': round(ips, 4), 'unit': 'images/s', 'cores': threads, 'kind': 'port', 'impl': 'torch-cpu', 'median_value': round(batch / med, 4), 'steps_timed': n,
            'sample': f'stock torch.nn CPU modules (same layer graph and weights; ATen/oneDNN kernels, limited to {threads} threads = cores), whole stage-{stage} step: '
                      f'forward, CE/MSE/FA, backward, torch.optim.SGD; B={batch} at {height}x{width}->{2 * height}x{2 * width}; {n} timed step(s) within a 12 s '
                      f'budget after one warm-up: best {dt:.2f} s (value), median {med:.2f} s'}


def cpu_baseline_numpy(state_dict, batch=1, height=256, width=512):
    """The numpy oracle (oracle/, the checker of the parity tests) on the host cores: whole model forward + losses + backward, B=1."""
    import numpy as np
    import oracle as O
    rs = np.random.RandomState(1234)
    sd = {k: v.detach().float().cpu().numpy() for k, v in state_dict.items() if 'num_batches' not in k}
    x = rs.standard_normal((batch, 3, height, width)).astype(np.float32)
    org = rs.standard_normal((batch, 3, 2 * height, 2 * width)).astype(np.float32)
    tg = rs.randint(0, 19, (batch, 2 * height, 2 * width)).astype(np.uint8); tg[rs.uniform(size=tg.shape) < 0.1] = 255
    cores = host_cores()

    def one_pass():
        t0 = time.time()
        out = O.model_forward(sd, x, 3, batch > 1)          # B=1: BatchNorm in eval mode (the global-pool BN needs B >= 2 to train, ASPP.py:39-40)
        O.total_loss(out, tg, org, 3)
        return time.time() - t0
    try:                                        # the BLAS behind numpy limited to the same core count as the torch leg; `cores` = what it then uses
        from threadpoolctl import threadpool_info, threadpool_limits
        with threadpool_limits(limits=cores, user_api='blas'):
            dt = one_pass()
            blas = [i.get('num_threads') for i in threadpool_info() if i.get('user_api') == 'blas']
        cores = max(blas) if blas else cores
    except ImportError:
        dt = one_pass()
    return {'value': round(batch / dt, 4), 'unit': 'images/s', 'cores': cores, 'kind': 'port', 'impl': 'numpy oracle',
            'sample': f'numpy oracle (fp32, multi-threaded BLAS), ResNet-101 + head forward, CE/MSE/FA, full backward, no optimizer update; '
                      f'B={batch} at {height}x{width}; one pass = {dt:.1f} s'}


# ----------------------------------------------------------------------------------------------------------------- kernel timers
def _time_ms(fn, reps, torch):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def decoder_stack_roofline(torch, HF, B, H, W, reps=10):
    """north_star: '>= 30 % of CDNA4 MFMA peak on the decoder conv stack'.  Every conv of the decoder head (ASPP, shortcut, cat_conv x2,
    cls_conv, SISR; DSRL.py:13-84) at the step's shapes, each pass launched on its own through the C ABI and timed with events on the
    launch stream; algorithmic (in-bounds) FLOP / time against the MFMA peak of the arithmetic the pass runs in."""
    from dualsuperreslearningforsemseg_amd import _lib
    h16, w16, h4, w4 = H // 16, W // 16, H // 4, W // 4
    shapes = [('aspp.0 1x1 2048->256', 2048, h16, w16, 256, 1, 0, 1), ('aspp.1 3x3 d6', 2048, h16, w16, 256, 3, 6, 6), ('aspp.2 3x3 d12', 2048, h16, w16, 256, 3, 12, 12),
              ('aspp.3 3x3 d18', 2048, h16, w16, 256, 3, 18, 18), ('aspp.5 1x1 1280->256', 1280, h16, w16, 256, 1, 0, 1), ('shortcut 1x1 256->48', 256, h4, w4, 48, 1, 0, 1),
              ('cat_conv.0 3x3 304->256', 304, h4, w4, 256, 3, 1, 1), ('cat_conv.4 3x3 256->256', 256, h4, w4, 256, 3, 1, 1), ('cls_conv 1x1 256->19', 256, h4, w4, 19, 1, 0, 1),
              ('SISR 3x3 304->192', 304, h4, w4, 192, 3, 1, 1)]
    mode = HF.get_conv_precision()
    arith = {'fp32': (0, 0, 0), 'bf16x3': (1, 1, 1), 'bf16x6': (2, 2, 2), 'mixed': (2, 1, 1), 'f16x3': (3, 3, 3), 'f16x1': (4, 4, 4)}[mode]       # forward, dgrad, wgrad
    tot = {'forward': [0.0, 0.0], 'dgrad': [0.0, 0.0], 'wgrad_per_layer': [0.0, 0.0], 'wgrad': [0.0, 0.0]}
    layers, keep, keep_probs = {}, [], []
    dev = torch.device('cuda', torch.cuda.current_device())
    for name, C, h, w, K, R, pad, dil in shapes:
        x = torch.randn((B, C, h, w), device=dev).contiguous(memory_format=torch.channels_last)
        wt = (torch.randn((K, C, R, R), device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
        Kp = (K + 3) & ~3
        dyb = torch.zeros((B, Kp, h, w), device=dev).contiguous(memory_format=torch.channels_last)
        dyb[:, :K].normal_()
        dy = dyb[:, :K]
        y = torch.empty((B, K, h, w), device=dev).contiguous(memory_format=torch.channels_last)
        dx, dw = torch.empty_like(x), torch.empty_like(wt)
        shp = (B, h, w, C, K, R, R, 1, pad, dil)
        st = HF._stream()
        wsf = HF._ws(HF.cquery('dsrl_conv2d_fwd_workspace_bytes', *shp), x)
        wsd = HF._ws(HF.cquery('dsrl_conv2d_dgrad_workspace_bytes', *shp), x)
        wsw = HF._ws(HF.cquery('dsrl_conv2d_wgrad_workspace_bytes', *shp), x)
        gf = 2.0 * HF.conv2d_inbounds_macs(*shp) / 1e9
        # operand magnitudes of the f16x3 arithmetic: in the step they are left by the producers of x / dy and by the per-step filter pass,
        # so they are measured once here, outside the brackets (the other arithmetics ignore them)
        xa = dya = wa = wsp = wtsp = None
        if mode == 'f16x3':
            wslot, wsp, wtsp, wtr = HF.split_filter(wt)         # the per-step filter pass of ddp.FlatParams, for this one filter
            keep.append((wslot, wsp, wtsp, wtr))
            xa, dya, wa, wsp, wtsp = HF.amax_for(x, x, C).data_ptr(), HF.amax_for(dy, dy, Kp).data_ptr(), wslot.data_ptr(), wsp.data_ptr(), wtsp.data_ptr()
        fwd_path = 'register-staged operands (conv_igemm_split_kernel)'
        if mode == 'f16x3' and HF.planes_wanted(x, C, C, K, R * R):
            # the production forward of this conv (functional._Conv2d, DSRL_PLANES_MODE=auto): both operands as fp16 planes staged by LDS-DMA.  The split pass
            # of the activation is part of its cost; cat_conv.0 and the SISR conv read the SAME concat buffer in the step (one pass for both: charged to cat_conv.0)
            wp_, wtp_ = HF.filter_planes(wt, wslot)
            xa_t = HF.amax_for(x, x, C)
            xp_ = HF.planes_of(x, C, xa_t)
            keep.append((wp_, wtp_, xp_))
            t_split = 0.0 if name.startswith('SISR') else _time_ms(lambda: HF.planes_of(x, C, xa_t), reps, torch)
            t_f = t_split + _time_ms(lambda: _lib.call('dsrl_conv2d_fwd_planes', x.data_ptr(), C, xa, xp_.data_ptr(), wt.data_ptr(), wa, wsp, wp_.data_ptr(), None, y.data_ptr(), K,
                                                       *shp, wsf.data_ptr(), wsf.numel(), None, 0, st), reps, torch)
            fwd_path = ('fp16-plane operands staged by LDS-DMA (conv_planes_kernel)' +
                        (f' + the split pass of its input ({t_split * 1e3:.0f} us)' if t_split else '; its input planes are the ones cat_conv.0 made (same concat buffer)'))
        else:
            t_f = _time_ms(lambda: _lib.call('dsrl_conv2d_fwd_amax', x.data_ptr(), C, xa, wt.data_ptr(), wa, wsp, None, y.data_ptr(), K, *shp, wsf.data_ptr(), wsf.numel(), None, 0, st), reps, torch)
        t_d = _time_ms(lambda: _lib.call('dsrl_conv2d_dgrad_amax', dy.data_ptr(), Kp, dya, wt.data_ptr(), None, wa, wtsp, dx.data_ptr(), C, *shp, wsd.data_ptr(), wsd.numel(),
                                         None, 0, None, 0, None, None, 0, None, 0, 0, st), reps, torch)
        t_w = _time_ms(lambda: _lib.call('dsrl_conv2d_wgrad_amax', x.data_ptr(), C, xa, dy.data_ptr(), Kp, dya, dw.data_ptr(), *shp, wsw.data_ptr(), wsw.numel(), st), reps, torch)
        layers[name] = {'gflop': round(gf, 2), 'forward_tflops': round(gf / t_f, 1), 'dgrad_tflops': round(gf / t_d, 1), 'wgrad_per_layer_tflops': round(gf / t_w, 1),
                        'forward_path': fwd_path}
        for k, t in (('forward', t_f), ('dgrad', t_d), ('wgrad_per_layer', t_w)):
            tot[k][0] += gf; tot[k][1] += t
        keep_probs.append((x, dy, dw, Kp, shp))
    # the production path launches the weight gradients of a backward pass as grouped grids (dsrl_conv2d_wgrad_group_*): the whole stack at once
    if mode != 'fp32':
        f16 = mode == 'f16x3'       # operand magnitudes: left by the producers in the step, measured once here
        amax = [(HF.amax_for(x, x, shp[3]), HF.amax_for(dy, dy, Kp)) if f16 else (None, None) for x, dy, dw, Kp, shp in keep_probs]

        def grouped():
            q = HF.WgradQueue()
            for (x, dy, dw, Kp, shp), (xa, dya) in zip(keep_probs, amax):
                q.add(x, shp[3], dy, Kp, dw, shp, None, xa, dya)
            q.flush()
        tot['wgrad'] = [tot['wgrad_per_layer'][0], _time_ms(grouped, reps, torch)]
    else:
        tot['wgrad'] = list(tot['wgrad_per_layer'])
    out = {'conv_arithmetic': mode, 'layers': layers,
           'note': 'each conv launched alone through the C ABI, on the path the training step takes for it (dgrad includes its own filter transpose), events on the launch stream, '
                   f'{reps} launches each; achieved = in-bounds FLOP / time; peak = dense bf16 MFMA 2516.6 TF / (3 | 6 MFMAs per product) or 157.3 TF fp32'}
    all_f = all_t = 0.0
    for i, k in enumerate(('forward', 'dgrad', 'wgrad')):
        peak = MFMA_PEAK_TFLOPS[arith[i]]
        ach = tot[k][0] / tot[k][1]
        out[k] = {'achieved': round(ach, 1), 'peak': round(peak, 1), 'frac': round(ach / peak, 4), 'arithmetic': ARITH_NAME[arith[i]], 'ms': round(tot[k][1], 3)}
        all_f += tot[k][0]; all_t += tot[k][1]
    out['wgrad']['how'] = 'all ten weight gradients of the stack as ONE grouped launch set (the production path; host table build included in the bracket)'
    out['wgrad_per_layer_launches'] = {'achieved': round(tot['wgrad_per_layer'][0] / tot['wgrad_per_layer'][1], 1), 'ms': round(tot['wgrad_per_layer'][1], 3),
                                       'frac': round(tot['wgrad_per_layer'][0] / tot['wgrad_per_layer'][1] / MFMA_PEAK_TFLOPS[arith[2]], 4)}
    out['all_passes'] = {'achieved': round(all_f / all_t, 1), 'ms': round(all_t, 3),
                         'frac_of_time_weighted_peak': round(sum(tot[k][1] * (tot[k][0] / tot[k][1]) / MFMA_PEAK_TFLOPS[arith[i]]
                                                                 for i, k in enumerate(('forward', 'dgrad', 'wgrad'))) / all_t, 4)}
    return out


def hbm_roofline(torch, HF, flat, B, H, W, reps=10):
    """The memory-bound kernels of the step, each launched on its own through the product wrappers: ALGORITHMIC bytes (every operand
    read / written once) / time against 8 TB/s."""
    dev = flat.device
    Ho, Wo, P = 2 * H, 2 * W, B * 4 * H * W
    rows = []

    def add(name, nbytes, ms, calls=1):
        gbs = nbytes / (ms * 1e-3) / 1e9
        rows.append({'kernel': name, 'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(gbs / HBM_PEAK_GBS, 4),
                     'algorithmic_bytes': int(nbytes), 'avg_launch_ms': round(ms, 4), 'launches_per_step': calls})

    from dualsuperreslearningforsemseg_amd import _lib
    logits = torch.randn((B, 19, Ho, Wo), device=dev).contiguous(memory_format=torch.channels_last)
    tgt = torch.randint(0, 19, (B, Ho, Wo), device=dev, dtype=torch.uint8)
    dl, out2, flag = torch.empty_like(logits), torch.empty(2, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
    ws = HF._ws(HF.cquery('dsrl_ce_fused_workspace_bytes', P), logits)
    st = HF._stream()
    add('ce_fused (CrossEntropy forward + d/dlogits + NaN check in one pass; with its count / finalize launches)', 2 * P * 19 * 4 + 2 * P,
        _time_ms(lambda: _lib.call('dsrl_ce_fused', logits.data_ptr(), 19, tgt.data_ptr(), P, 19, 255, dl.data_ptr(), 19, out2.data_ptr(), flag.data_ptr(),
                                   ws.data_ptr(), ws.numel(), st), reps, torch))
    a = torch.randn((B, 3, Ho, Wo), device=dev).contiguous(memory_format=torch.channels_last)
    b = torch.randn((B, 3, Ho, Wo), device=dev).contiguous(memory_format=torch.channels_last)
    da = torch.empty_like(a)
    ws2 = HF._ws(HF.cquery('dsrl_mse_workspace_bytes', a.numel()), a)
    add('mse_fused (MSE forward + gradient + NaN check in one pass)', 3 * P * 3 * 4,
        _time_ms(lambda: _lib.call('dsrl_mse_fused', a.data_ptr(), b.data_ptr(), a.numel(), 0.1, da.data_ptr(), out2.data_ptr(), flag.data_ptr(), ws2.data_ptr(), ws2.numel(), st), reps, torch))
    x = torch.randn((B, 19, H, W), device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wt = torch.randn((19, 19, 2, 2), device=dev).requires_grad_(True)
    bias = torch.zeros(19, device=dev, requires_grad=True)
    y = HF.conv_transpose2d_k2s2(x, wt, bias)
    dy = torch.randn_like(y)
    add('convt2x2_fwd (final ConvTranspose2d 19->19, logits write)', (P // 4 + P) * 19 * 4, _time_ms(lambda: HF.conv_transpose2d_k2s2(x.detach(), wt.detach(), bias.detach()), reps, torch))
    add('convt2x2_bwd (dx + dw + db)', (P + 2 * P // 4) * 19 * 4, _time_ms(lambda: torch.autograd.grad(y, (x, wt, bias), dy, retain_graph=True), reps, torch))
    # the production tail when the shape qualifies (round 5): the CE value inside the forward kernel, its gradient formed inside the backward kernel
    xd, yd = x.detach(), torch.empty_like(y)
    if HF.cquery('dsrl_convt2x2_fwd_ce_supported', xd.data_ptr(), yd.data_ptr(), B, H, W, 19, 19) and \
            HF.cquery('dsrl_convt2x2_bwd_ce_supported', xd.data_ptr(), yd.data_ptr(), tgt.data_ptr(), B, H, W, 19, 19):
        scal = torch.zeros(8, device=dev)
        wsf = HF._ws(HF.cquery('dsrl_convt2x2_fwd_ce_workspace_bytes', B, H, W), xd)
        add('convt2x2_fwd_ce (final ConvTranspose2d forward + CrossEntropy value + NaN check in one kernel; with its finalize launch)', (P // 4 + P) * 19 * 4 + P,
            _time_ms(lambda: _lib.call('dsrl_convt2x2_fwd_ce', xd.data_ptr(), wt.data_ptr(), bias.data_ptr(), yd.data_ptr(), B, H, W, 19, 19, tgt.data_ptr(), 255,
                                       scal.data_ptr(), flag.data_ptr(), wsf.data_ptr(), wsf.numel(), st), reps, torch))
        dxb, dwb, dbb = torch.empty_like(xd), torch.empty_like(wt), torch.empty(19, device=dev)
        wsb = HF._ws(HF.cquery('dsrl_convt2x2_bwd_workspace_bytes', B, H, W, 19, 19), xd)
        ftg, ftw = torch.randn((B, Ho // 8, Wo // 8), device=dev), torch.randn(19, device=dev)
        add('convt2x2_bwd_ce (d(CE)/d(logits) formed inside + feature-transformer term + dx + dw + db; replaces ce_fused\'s gradient write and convt2x2_bwd\'s read)',
            (P + 2 * P // 4) * 19 * 4 + P,
            _time_ms(lambda: _lib.call('dsrl_convt2x2_bwd_ce', xd.data_ptr(), wt.data_ptr(), yd.data_ptr(), tgt.data_ptr(), 255, scal.data_ptr() + 4, ftg.data_ptr(), ftw.data_ptr(), 8,
                                       dxb.data_ptr(), dwb.data_ptr(), dbb.data_ptr(), B, H, W, 19, 19, wsb.data_ptr(), wsb.numel(), st), reps, torch))
    bn = torch.nn.BatchNorm2d(256).to(dev).train()
    xb = torch.randn((B, 256, H // 4, W // 4), device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    nb = xb.numel() * 4
    yb = HF.batch_norm_act(xb, bn, relu=True)
    dyb = torch.randn_like(yb)
    add('bn large tensor forward (layer1 bn3 shape, statistics + apply + ReLU)', 2 * nb, _time_ms(lambda: HF.batch_norm_act(xb.detach(), bn, relu=True), reps, torch), 8)
    add('bn large tensor backward', 4 * nb, _time_ms(lambda: torch.autograd.grad(yb, xb, dyb, retain_graph=True), reps, torch), 8)
    g0, p0, m0 = flat.g_flat.clone(), flat.p_flat.clone(), flat.m_flat.clone()
    add('sgd_kernel (p, g, momentum arenas)', 5 * flat.numel * 4, _time_ms(lambda: HF.sgd_step_(flat.p_flat, flat.g_flat, flat.m_flat, 0.0, 0.9, 0.0, 1.0), reps, torch))
    flat.g_flat.copy_(g0); flat.p_flat.copy_(p0); flat.m_flat.copy_(m0)
    return rows


# ----------------------------------------------------------------------------------------------------------------- main
def main():
    args = parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args))

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if os.environ.get('DSRL_ALL_RANKS_ON_GPU0'):                       # rehearsal of the multi-rank path on a one-GPU box
        local = 0
        os.environ['DSRL_BN_FUSED'] = '0'     # several processes on one GPU: the fused BN kernels' device-wide barrier needs the GPU to itself
    if args.gpus != world:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or without torch.distributed.run)')
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
    from dualsuperreslearningforsemseg_amd import functional as HF
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
    weights0 = None if (args.no_cpu_baseline or world > 1) else {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}

    host_ms = []
    batch_now = [img, org, tgt]             # the batch run() trains on (the config-5 figure swaps in a 512x1024 one)

    def run(n):
        # train_or_resume()'s loop: iteration k's loss/NaN readback is collected after iteration k+1 has been enqueued
        last = None
        for _ in range(n):
            step.enqueue(batch_now[0], batch_now[1], batch_now[2], hp['lr'], hp['momentum'], hp['weight_decay'], True)
            host_ms.append(step.host_enqueue_s * 1e3)
            while step.pending() > 1:
                last = step.collect()
        while step.pending():
            last = step.collect()
        return last

    def timed(nwarm, nsteps):
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides; MAX over ranks."""
        run(nwarm)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        del host_ms[:]
        del step.comm_events[:]
        t0 = time.perf_counter()
        last = run(nsteps)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt)
        return el, last

    step.time_collectives = world > 1
    elapsed, losses = timed(args.warmup, args.steps)
    host_enqueue = sorted(host_ms)[len(host_ms) // 2] if host_ms else 0.0
    replays = step.graph_replays
    comm = None
    if world > 1:
        # the collectives of the timed region, from HIP events on the compute stream (rank 0's view): the rank-0 BatchNorm-buffer broadcast in front
        # of the replay, and the part of the gradient all-reduce that is NOT hidden behind the second backward graph (end of that graph ->
        # all collectives complete); plus the ADVICE check that no fused BatchNorm launch gave up at its device-wide barrier beside RCCL
        evs = list(step.comm_events)
        step.time_collectives = False
        section = sum(e[2].elapsed_time(e[3]) for e in evs) / max(len(evs), 1)
        comm = {'rccl_world_size': dist.get_world_size(), 'backend': dist.get_backend(),
                'broadcast_ms_per_step': round(sum(e[0].elapsed_time(e[1]) for e in evs) / max(len(evs), 1), 4),
                'steps_bracketed': len(evs), 'gradient_bytes_per_step': int(flat.numel * 4)}
        if step.split:
            comm['allreduce_exposed_ms_per_step'] = round(section, 4)
            comm['schedule'] = ('two hipGraphs per step: the all-reduce of the chunks complete after the first (head, ASPP, layer4) runs beside the second (layers 3..1), '
                                'the rest behind it')
        else:
            # one-graph schedule: the bracket holds the chunked all-reduce AND the SGD kernels that run under it; the optimiser pass alone is timed on
            # scratch arenas of the same size, and what the exchange adds to the step is the difference
            scratch = [torch.zeros(flat.numel, device=dev) for _ in range(3)]
            hyp = torch.tensor([0.0, 0.9, 5e-4, 1.0], device=dev)
            for _ in range(3):
                HF.sgd_step_dev_(scratch[0], scratch[1], scratch[2], hyp)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                HF.sgd_step_dev_(scratch[0], scratch[1], scratch[2], hyp)
            e1.record(); torch.cuda.synchronize()
            sgd_ms = e0.elapsed_time(e1) / 10
            del scratch
            exposed = max(section - sgd_ms, 1e-6)
            comm.update({'exchange_and_update_ms_per_step': round(section, 4), 'sgd_alone_ms': round(sgd_ms, 4),
                         'allreduce_exposed_ms_per_step': round(exposed, 4),
                         'allreduce_algorithm_bandwidth_GBps': round(flat.numel * 4 / exposed / 1e6, 1),
                         'reduce_chunks': int(os.environ.get('DSRL_REDUCE_CHUNKS', '4')),
                         'schedule': 'one hipGraph per step; behind it the gradient arena is all-reduced in `reduce_chunks` ranges and the SGD kernel of a range runs under '
                                     'the all-reduce of the next ones'})
        stuck = HF.bn_fused_barrier_timeouts()
        assert stuck == 0, f'{stuck} fused BatchNorm blocks timed out at their device-wide barrier while RCCL shared the device'
        comm['bn_fused_barrier_timeouts'] = stuck
    if world > 1:
        # data-parallel invariant: after K identical-seed steps every rank must hold bit-identical parameters
        chk = torch.stack([flat.p_flat.double().sum(), flat.p_flat.double().abs().sum()])
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert torch.equal(lo, hi), f'ranks diverged: {lo.tolist()} vs {hi.tolist()}'

    # the same step under the other conv arithmetics, timed over the SAME warm-up / step counts (every rank takes part)
    by_arith = None
    default_mode = HF.get_conv_precision()
    if not args.no_prof:
        gb = args.batch * world
        by_arith = {default_mode: round(gb * args.steps / elapsed, 1)}
        for mode in ('f16x3', 'f16x1', 'bf16x6', 'mixed', 'fp32'):
            if mode == default_mode or (mode == 'fp32' and world > 1):        # exact-product fp32 MFMA: single-GPU figure only
                continue
            HF.set_conv_precision(mode)
            el, _ = timed(args.warmup, args.steps)
            by_arith[mode] = round(gb * args.steps / el, 1)
        HF.set_conv_precision(None)

    # BASELINE config 5's size (512x1024 -> 1024x2048, per-GPU batch 8) as a bounded extra of the default line: 4 untimed + 8 timed steps
    config5 = None
    if not args.no_config5 and world == 1 and (args.stage, args.height, args.width, args.batch) == (3, 256, 512, 8):
        (img5, org5), (tgt5, _) = next(iter(SyntheticCityscapes(args.batch, (512, 1024), dev, rank=rank, length=1)))
        batch_now[:] = [img5, org5, tgt5]
        config5 = {'workload': 'the same stage-3 step at 512x1024 input -> 1024x2048 logits, per-GPU batch 8 (BASELINE.json configs[4] on one GPU), 4 untimed + 8 timed steps',
                   'steps': 8, 'images_per_s_by_conv_arithmetic': {}}
        for mode in (default_mode, 'f16x1'):            # f16x1: the reduced-precision arithmetic that configuration names ('fp16 MFMA convs' = apex O1 / O2: one fp16 MMA per product)
            HF.set_conv_precision(mode)
            el5, l5 = timed(4, 8)
            config5['images_per_s_by_conv_arithmetic'][mode] = round(args.batch * 8 / el5, 1)
            if mode == default_mode:
                config5.update(value=round(args.batch * 8 / el5, 2), unit='images/s', ms_per_step=round(1e3 * el5 / 8, 2), losses_last_step=[round(v, 5) for v in l5])
        HF.set_conv_precision(None)
        batch_now[:] = [img, org, tgt]
        del img5, org5, tgt5
        run(3); torch.cuda.synchronize()            # back on the headline shape (its graph is still cached) before the per-kernel passes

    def read_prof(nsteps, stride=1):
        fams = []
        for fam in range(15):                # family = 3 * arithmetic + pass (include/dsrl_hip.h)
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

    roof = None
    if not args.no_prof:
        # Per-kernel durations: the timed region replays a hipGraph (no place for per-launch events) and overlaps weight-gradient kernels
        # with the rest of backward, so the roofline comes from a few extra EAGER steps right behind it, every conv launch bracketed by
        # HIP events on its launch stream and the weight-gradient side stream off (a kernel's bracket is then its own execution).
        graph_was, overlap_was = step.use_graph, HF.overlap_wgrad
        step.use_graph, HF.overlap_wgrad = False, False
        run(2); torch.cuda.synchronize()
        lib.dsrl_prof_enable(1)
        excl_steps = 5
        run(excl_steps); torch.cuda.synchronize()
        roof = {'bound': 'mfma', 'unit': 'TFLOP/s', 'traffic': None}
        roof.update(read_prof(excl_steps))
        roof['measured'] = (f'{excl_steps} eager steps right after the timed region (same process, same tensors), HIP events around every conv launch on its '
                            'launch stream, weight-gradient side stream disabled (exclusive kernel execution)')
        step.use_graph, HF.overlap_wgrad = graph_was, overlap_was
        try:        # HBM-side bytes per launch from the committed rocprofv3 --pmc passes of this same command (not collectable in-process)
            import hashlib
            src_now = hashlib.sha256(open(os.path.join(ROOT, 'dualsuperreslearningforsemseg_amd', 'csrc', 'conv_igemm.hip'), 'rb').read()).hexdigest()[:16]
            for fn in ('round5_pmc_traffic.json', 'round4_pmc_traffic.json', 'round3_pmc_traffic.json', 'round2_pmc_traffic.json', 'round1_pmc_traffic.json'):
                path = os.path.join(ROOT, 'profiles', fn)
                if not os.path.isfile(path):
                    continue
                pmc = json.load(open(path))
                if (args.stage, args.height, args.width, args.batch) == (3, 256, 512, 8) and roof.get('kernel') in pmc:
                    roof['traffic'] = pmc[roof['kernel']]['hbm_bytes_per_launch']
                    stamp = pmc.get('_stamp', {})
                    roof['traffic_source'] = (f'profiles/{fn} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH_SIZE x2 on gfx950), taken at commit '
                                              f"{stamp.get('commit', 'unknown')} with conv_igemm.hip sha256 {stamp.get('conv_igemm_sha16', 'unknown')}")
                    pl_now = hashlib.sha256(open(os.path.join(ROOT, 'dualsuperreslearningforsemseg_amd', 'csrc', 'conv_planes.hip'), 'rb').read()).hexdigest()[:16]
                    # false: the kernels changed since that profile (profiles before round 4 carry no stamp for conv_planes.hip)
                    w3_now = hashlib.sha256(open(os.path.join(ROOT, 'dualsuperreslearningforsemseg_amd', 'csrc', 'conv_wgrad3.hip'), 'rb').read()).hexdigest()[:16]
                    # round 5: the forward / dgrad kernel template lives in conv_split_kernel.h, its 128x128 two-group builds in conv_sk.hip
                    hk_now = hashlib.sha256(open(os.path.join(ROOT, 'dualsuperreslearningforsemseg_amd', 'csrc', 'conv_split_kernel.h'), 'rb').read() +
                                            open(os.path.join(ROOT, 'dualsuperreslearningforsemseg_amd', 'csrc', 'conv_sk.hip'), 'rb').read()).hexdigest()[:16]
                    roof['traffic_kernel_source_unchanged'] = (stamp.get('conv_igemm_sha16') == src_now and stamp.get('conv_planes_sha16', pl_now) == pl_now
                                                               and stamp.get('conv_wgrad3_sha16', w3_now) == w3_now and stamp.get('conv_split_kernel_sha16') == hk_now)
                    break
            # matrix-pipe busy fraction of the dominant family from the committed SQ-counter pass of this command (tools/pmc_step.sh): the share of
            # SIMD cycles in which an MFMA executes - what `frac` leaves unsaid about WHY a launch is below its roof
            sq_name = next((n for n in ('round5_sq_counters_step.json', 'round4_sq_counters_step.json') if os.path.isfile(os.path.join(ROOT, 'profiles', n))), 'round5_sq_counters_step.json')
            sq_path = os.path.join(ROOT, 'profiles', sq_name)
            if os.path.isfile(sq_path):
                fam = json.load(open(sq_path)).get('_families', {}).get(roof.get('kernel'))
                if fam:
                    roof['mfma_busy'] = fam['mfma_busy']
                    roof['mfma_busy_source'] = f'profiles/{sq_name} (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE over the graph-replayed step)'
        except Exception:
            pass
        if world == 1:
            roof['decoder_stack'] = decoder_stack_roofline(torch, HF, args.batch, args.height, args.width)

    hbm = None
    if not args.no_prof and world == 1:
        hbm = hbm_roofline(torch, HF, flat, args.batch, args.height, args.width)

    if rank == 0:
        gb = args.batch * world
        is_headline = (args.stage, args.height, args.width) == (3, 256, 512)
        line = {
            'metric': 'stage-3 train images/sec at 256x512->512x1024' if is_headline
                      else f'stage-{args.stage} train images/sec at {args.height}x{args.width}->{2 * args.height}x{2 * args.width}',
            'value': round(gb * args.steps / elapsed, 3), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * elapsed / args.steps, 3), 'host_enqueue_ms_per_step': round(host_enqueue, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'conv_arithmetic': ARITH_TEXT[default_mode], 'images_per_s_by_conv_arithmetic': by_arith, 'data': 'synthetic',
            'config': {'workload': f'DSRL stage {args.stage} (ResNet-101 OS16 + ASPP + SSSR/SISR decoders + FA loss), full train step, '
                                   f'random-init weights, {args.height}x{args.width} input -> {2 * args.height}x{2 * args.width} logits'
                                   + ('' if is_headline else ' (not the headline configuration; BASELINE config 5 is --height 512 --width 1024)'),
                       'global_batch': gb, 'per_gpu_batch': args.batch, 'parallelism': f'dp{world}', 'optimizer': 'SGD m0.9 wd5e-4 lr0.006',
                       'step_execution': (f'hipGraph replay ({replays} of the {args.steps + args.warmup} headline iterations; collectives and SGD follow the replay when dp > 1)'
                                          if replays else 'eager launches'),
                       'losses_last_step': [round(v, 5) for v in losses]},
            'roofline': roof, 'roofline_hbm': hbm,
        }
        if comm is not None:
            line['collectives'] = comm
        if config5 is not None:
            line['config5'] = config5
        if weights0 is not None:
            line['cpu_baseline'] = cpu_baseline_torch(weights0, args.stage)
            # BASELINE.json configs[0]: the reference's own CPU-runnable case - stage 1 (SSSR only), batch 2, 128x256 input, --device cpu
            c1 = cpu_baseline_torch(weights0, 1, batch=2, height=128, width=256)
            c1['config'] = 'BASELINE.json configs[0]: stage-1 SSSR, batch 2, 128x256 input, --device cpu (train_stage1_cmdline.json)'
            line['cpu_baseline_c1'] = c1
            if args.stage == 3:
                line['cpu_baseline_numpy_port'] = cpu_baseline_numpy(weights0)
        print(json.dumps(line), flush=True)
    step.release()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
