import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dualsuperreslearningforsemseg_amd import functional as HF
dev = 'cuda:0'
rs = np.random.RandomState(0)
for shape in ((8, 256, 16, 32, 256, 3, 1, 1, 1), (2, 256, 8, 16, 256, 3, 1, 1, 1), (8, 256, 16, 32, 1024, 1, 1, 0, 1), (2, 304, 64, 128, 256, 3, 1, 1, 1), (2, 512, 8, 16, 512, 3, 1, 2, 2)):
    N, C, H, W, K, R, stride, pad, dil = shape
    x = torch.from_numpy(np.maximum(rs.standard_normal((N, C, H, W)), 0).astype(np.float32) + 3.0).to(dev).contiguous(memory_format=torch.channels_last)
    w = torch.from_numpy((rs.standard_normal((K, C, R, R)) / np.sqrt(C * R * R)).astype(np.float32)).to(dev).contiguous(memory_format=torch.channels_last)
    rec, wsp, wtsp, wtr = HF.split_filter(w)
    xa = HF.amax_for(x)
    shp = (N, H, W, C, K, R, R, stride, pad, dil)
    parts = int(HF.query('dsrl_conv2d_fwd_stats_parts', *shp))
    Ho, Wo = (H + 2 * pad - dil * (R - 1) - 1) // stride + 1, (W + 2 * pad - dil * (R - 1) - 1) // stride + 1
    y = torch.empty((N, K, Ho, Wo), device=dev).contiguous(memory_format=torch.channels_last)
    ws = torch.empty(int(HF.query('dsrl_conv2d_fwd_workspace_bytes', *shp)) + 4096, device=dev, dtype=torch.uint8)
    stats = torch.zeros(int(HF.query('dsrl_bn_stats_floats', 3, max(parts, 1), K)), device=dev)
    HF.call('dsrl_conv2d_fwd_planes', x.data_ptr(), C, xa.data_ptr(), None, w.data_ptr(), rec.data_ptr(), wsp.data_ptr(), None, None, y.data_ptr(), K, *shp,
            ws.data_ptr(), ws.numel(), stats.data_ptr(), parts, HF._stream())
    torch.cuda.synchronize()
    pt = stats[:3 * parts * K].cpu().numpy().reshape(3, parts, K).astype(np.float64)
    n = pt[0].sum(0); mu = (pt[0] * pt[1]).sum(0) / n
    m2 = (pt[2] + pt[0] * (pt[1] - mu) ** 2).sum(0)
    yy = y.cpu().numpy().astype(np.float64).transpose(0, 2, 3, 1).reshape(-1, K)
    em = np.abs(mu - yy.mean(0)).max() / np.abs(yy.mean(0)).max()
    ev = np.abs(m2 / n - yy.var(0)).max() / yy.var(0).max()
    print(shape, 'parts', parts, 'mean err %.2e var err %.2e' % (em, ev), 'n ok', bool(np.all(n == yy.shape[0])))
