#!/usr/bin/env python3
"""fp32 / bf16x3 / bf16x6 / f16x3 conv kernels: timing per shape and error against an fp64 torch reference (raw C ABI).
PREC_DY=tiny scales dy to ~1e-7 with a heavy (log-normal) tail, the way gradients of a mean-reduced loss look."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dualsuperreslearningforsemseg_amd import _lib
from sweep_conv import SHAPES, t_ms
lib = _lib.load()
SHAPES['l4_3x3'] = (8, 512, 16, 32, 512, 3, 1, 2, 2)
SHAPES['aspp_d6'] = (8, 2048, 16, 32, 256, 3, 1, 6, 6)
SHAPES['sisr'] = (8, 304, 64, 128, 192, 3, 1, 1, 1)
SHAPES['cls'] = (8, 256, 64, 128, 19, 1, 1, 0, 1)
dev = 'cuda:0'
for name, (N, C, H, W, K, R, stride, pad, dil) in SHAPES.items():
    Ho = (H + 2 * pad - dil * (R - 1) - 1) // stride + 1; Wo = (W + 2 * pad - dil * (R - 1) - 1) // stride + 1
    torch.manual_seed(0)
    x = torch.relu(torch.randn(N * H * W * C, device=dev)); w = torch.randn(K * R * R * C, device=dev) * (2.0 / (C * R * R)) ** 0.5
    dy = torch.randn(N * Ho * Wo * K, device=dev)
    if os.environ.get('PREC_DY') == 'tiny':
        dy = dy * 1e-7 * torch.exp(2.0 * torch.randn_like(dy))
    shp = (N, H, W, C, K, R, R, stride, pad, dil)
    ws = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    gf = 2 * lib.dsrl_conv2d_inbounds_macs(*shp) / 1e9
    res = {}
    ref64 = {}
    if gf < 12:      # fp64 references on the host for the moderate shapes
        xc = x.view(N, H, W, C).permute(0, 3, 1, 2).double().cpu().requires_grad_(True)
        wc = w.view(K, R, R, C).permute(0, 3, 1, 2).double().cpu().requires_grad_(True)
        yc = torch.nn.functional.conv2d(xc, wc, None, stride, pad, dil)
        yc.backward(dy.view(N, Ho, Wo, K).permute(0, 3, 1, 2).double().cpu())
        ref64 = {'fwd': yc.detach().permute(0, 2, 3, 1).reshape(-1), 'dgrad': xc.grad.permute(0, 2, 3, 1).reshape(-1), 'wgrad': wc.grad.permute(0, 2, 3, 1).reshape(-1)}
    for what in ('fwd', 'dgrad', 'wgrad'):
        outs, times = [], []
        for prec in (0, 1, 2, 4):
            lib.dsrl_conv_precision(prec)
            if what == 'fwd':
                o = torch.empty(N * Ho * Wo * K, device=dev)
                f = lambda: _lib.check(lib.dsrl_conv2d_fwd(x.data_ptr(), C, w.data_ptr(), None, o.data_ptr(), K, *shp, ws.data_ptr(), ws.numel(), st), 'fwd')
            elif what == 'dgrad':
                o = torch.empty(N * H * W * C, device=dev)
                if K % 4: break
                f = lambda: _lib.check(lib.dsrl_conv2d_dgrad(dy.data_ptr(), K, w.data_ptr(), None, o.data_ptr(), C, *shp, ws.data_ptr(), ws.numel(), st), 'dgrad')
            else:
                o = torch.empty(K * R * R * C, device=dev)
                if K % 4: break
                f = lambda: _lib.check(lib.dsrl_conv2d_wgrad(x.data_ptr(), C, dy.data_ptr(), K, o.data_ptr(), *shp, ws.data_ptr(), ws.numel(), st), 'wgrad')
            times.append(t_ms(f, 10)); outs.append(o.clone())
        if len(outs) == 4:
            if what in ref64:
                r = ref64[what]; errs = [float((o.double().cpu() - r).abs().max() / r.abs().max()) for o in outs]; tag = 'err vs fp64'
            else:
                errs = [float((o - outs[0]).abs().max() / outs[0].abs().max()) for o in outs]; tag = 'err vs fp32'
            res[what] = (f'{what}: fp32 {times[0]*1e3:.0f}us/{gf/times[0]:.0f}TF x3 {times[1]*1e3:.0f}us/{gf/times[1]:.0f}TF x6 {times[2]*1e3:.0f}us/{gf/times[2]:.0f}TF f16x3 {times[3]*1e3:.0f}us/{gf/times[3]:.0f}TF '
                         f'{tag} {errs[0]:.1e}/{errs[1]:.1e}/{errs[2]:.1e}/{errs[3]:.1e}')
    lib.dsrl_conv_precision(0)
    print(f'{name:10s} ' + ' | '.join(res.values()), flush=True)
