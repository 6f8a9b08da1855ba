#!/usr/bin/env python3
"""What would running the SISR branch beside the SSSR decoder's tail buy?  The two are independent between the concat and the losses (DSRL.py:168-177):
the SISR decoder is one large MFMA-bound conv (+ PixelShuffle), the SSSR tail a chain of memory-bound kernels (cls_conv, bilinear x2, dropout, two
ConvTranspose, a 19-channel BatchNorm).  Forward only, each variant captured into a hipGraph and replayed (HIP events): serial on one stream vs the SISR branch on a second stream.
Not a product path: it only measures how much of the shorter chain the hardware hides (workspaces are shared between the streams here, results are not checked)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dualsuperreslearningforsemseg_amd as D
from dualsuperreslearningforsemseg_amd import functional as HF
from dualsuperreslearningforsemseg_amd.datasets.Cityscapes import settings as cs
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = D.DSRL(3, cs).to(dev).to(memory_format=torch.channels_last).train()
from dualsuperreslearningforsemseg_amd.ddp import FlatParams
flat = FlatParams(model)                # the step's filter pass: pre-split filters and filter planes, so that the convs take their production kernels


def prep():
    HF.amax_begin_step(dev); flat.refresh_transposed_filters()
B = 8
cat = torch.randn(B, 304, 128, 256, device=dev).contiguous(memory_format=torch.channels_last)
mid = torch.randn(B, 256, 128, 256, device=dev).contiguous(memory_format=torch.channels_last)
sd = model.SSSR_decoder
side = torch.cuda.Stream()


def sssr_tail():
    with torch.no_grad():
        return sd['upsample16_pred'](sd['cls_conv'](mid))


def sssr_whole():
    with torch.no_grad():
        return sd['upsample16_pred'](sd['cls_conv'](sd['cat_conv'](cat)))


def sisr():
    with torch.no_grad():
        return model.SISR_decoder(cat)


def timeit(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def both(first):
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        y2 = sisr()
    y1 = first()
    cur.wait_stream(side)
    return y1, y2


def graphed(f):
    """f captured into a hipGraph (the step's execution mode: no host gaps between the kernels); returns a replay callable"""
    for _ in range(3):
        prep(); f()
    prep()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    HF.graph_keepalive = []
    try:
        with torch.cuda.graph(g):
            out = f()
    finally:
        keep, HF.graph_keepalive = HF.graph_keepalive, None
    g._keep = (keep, out)
    return g.replay


HF.begin_forward(True)
for name, f in (('SSSR tail (cls_conv .. last ConvTranspose)', sssr_tail), ('SSSR decoder (cat_conv x2 + tail)', sssr_whole)):
    g1, g2 = graphed(f), graphed(sisr)
    gs = graphed(lambda: (f(), sisr()))
    gc = graphed(lambda: both(f))
    t1, t2, ts, tc = timeit(g1, 20), timeit(g2, 20), timeit(gs, 20), timeit(gc, 20)
    print(f'{name}: {t1:.0f} us, SISR decoder {t2:.0f} us; one after the other {ts:.0f} us, SISR on a second stream {tc:.0f} us  (hidden: {ts - tc:.0f} us)', flush=True)
