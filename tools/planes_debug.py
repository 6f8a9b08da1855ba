#!/usr/bin/env python3
"""Diagnosis of the planes path on one small conv: are the planes what the split should give, and which of them does the kernel actually read?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualsuperreslearningforsemseg_amd import functional as HF  # noqa: E402
from dualsuperreslearningforsemseg_amd._lib import call, query  # noqa: E402

dev = 'cuda:0'
torch.manual_seed(0)
N, C, H, W, K, R, stride, pad, dil = 2, 64, 16, 32, 64, 1, 1, 0, 1
x = torch.randn((N, C, H, W), device=dev).contiguous(memory_format=torch.channels_last)
w = (torch.randn((K, C, R, R), device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
shp = (N, H, W, C, K, R, R, stride, pad, dil)
rec, wsp, wtsp, wt = HF.split_filter(w)
wp, wtp = HF.filter_planes(w, rec)
xa = HF.amax_slot(x.device); xa.zero_()
call('dsrl_amax', x.data_ptr(), C, N * H * W, C, xa.data_ptr(), HF._stream())
xp = HF.planes_of(x, C, xa)
torch.cuda.synchronize()
P = N * H * W
lo_off = int(query('dsrl_planes_lo_offset', P * C))
amax = xa.view(torch.float32).abs().max().item() if False else None
bits = xa.cpu().numpy().view('uint32').max()
import numpy as np
am = np.array([bits], dtype=np.uint32).view(np.float32)[0]
ex = int((bits >> 23) & 0xff); sh = 14 - (ex - 127)
print('amax', am, 'shift', sh)
xs = torch.ldexp(x.permute(0, 2, 3, 1).reshape(P, C), torch.tensor(sh, device=dev))
hi_ref = xs.half(); lo_ref = (xs - hi_ref.float()).half()
hi = xp[:P * C * 2].view(torch.float16).view(P, C); lo = xp[lo_off:lo_off + P * C * 2].view(torch.float16).view(P, C)
print('split hi equal', torch.equal(hi, hi_ref), 'lo equal', torch.equal(lo, lo_ref), 'lo absmax', lo.float().abs().max().item())
wlo_off = int(query('dsrl_planes_lo_offset', K * C * R * R))
wbits = rec.cpu().numpy().view('uint32').max(); wex = int((wbits >> 23) & 0xff); wsh = 14 - (wex - 127)
wsv = torch.ldexp(w.permute(0, 2, 3, 1).reshape(K, -1), torch.tensor(wsh, device=dev))
whi_ref = wsv.half(); wlo_ref = (wsv - whi_ref.float()).half()
whi = wp[:K * C * R * R * 2].view(torch.float16).view(K, -1); wlo = wp[wlo_off:wlo_off + K * C * R * R * 2].view(torch.float16).view(K, -1)
print('filter hi equal', torch.equal(whi, whi_ref), 'lo equal', torch.equal(wlo, wlo_ref))
ws = HF._ws(HF.cquery('dsrl_conv2d_fwd_workspace_bytes', *shp), x)
st = HF._stream()


def fwd(xpl, wpl):
    y = HF.new_cl((N, K, H, W), x)
    call('dsrl_conv2d_fwd_planes', x.data_ptr(), C, xa.data_ptr(), None if xpl is None else xpl.data_ptr(), w.data_ptr(), rec.data_ptr(), wsp.data_ptr(),
         None if wpl is None else wpl.data_ptr(), None, y.data_ptr(), K, *shp, ws.data_ptr(), ws.numel(), None, 0, st)
    torch.cuda.synchronize()
    return y


y_ref = fwd(None, None)
y_pl = fwd(xp, wp)
print('planes vs igemm: equal', torch.equal(y_ref, y_pl), 'rel', ((y_ref - y_pl).abs().max() / y_ref.abs().max()).item())
# expectation from the planes themselves, in fp64
hi64, lo64, whi64, wlo64 = hi.double(), lo.double(), whi.double(), wlo.double()
full = (hi64 @ whi64.T + hi64 @ wlo64.T + lo64 @ whi64.T) * 2.0 ** (-(sh + wsh))
hh = (hi64 @ whi64.T) * 2.0 ** (-(sh + wsh))
yp = y_pl.permute(0, 2, 3, 1).reshape(P, K).double(); yr = y_ref.permute(0, 2, 3, 1).reshape(P, K).double()
sc = full.abs().max()
print('planes kernel vs full 3-term', ((yp - full).abs().max() / sc).item(), 'vs hi*hi only', ((yp - hh).abs().max() / sc).item())
print('igemm  kernel vs full 3-term', ((yr - full).abs().max() / sc).item(), 'vs hi*hi only', ((yr - hh).abs().max() / sc).item())
# which planes does the kernel read?  zero the second plane of x / of w in a copy
xp0 = xp.clone(); xp0[lo_off:] = 0
wp0 = wp.clone(); wp0[wlo_off:] = 0
print('x lo zeroed changes result:', not torch.equal(fwd(xp0, wp), y_pl), ' w lo zeroed changes result:', not torch.equal(fwd(xp, wp0), y_pl))
xph = xp.clone(); xph[:P * C * 2] = 0
print('x hi zeroed: max |y| =', fwd(xph, wp).abs().max().item(), ' (full', y_pl.abs().max().item(), ')')
d = (wlo.float() - wlo_ref.float()).abs()
i = int(d.argmax())
print('filter lo: max diff', d.max().item(), 'at', divmod(i, wlo.shape[1]), 'got', wlo.flatten()[i].item(), 'ref', wlo_ref.flatten()[i].item(), 'hi', whi.flatten()[i].item(),
      'mismatches', int((wlo != wlo_ref).sum()), 'of', wlo.numel())
wtlo = wtp[wlo_off:wlo_off + K * C * R * R * 2].view(torch.float16).view(C, -1)
wthi = wtp[:K * C * R * R * 2].view(torch.float16).view(C, -1)
print('transposed planes: hi equal', torch.equal(wthi, whi_ref.view(K, R * R, C).permute(2, 1, 0).reshape(C, -1)), 'lo equal', torch.equal(wtlo, wlo_ref.view(K, R * R, C).permute(2, 1, 0).reshape(C, -1)))
