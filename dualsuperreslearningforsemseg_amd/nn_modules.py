"""torch.nn module classes whose forward runs on libdsrl_hip.so.

They subclass the stock torch classes (same constructor arguments, parameter names and state_dict keys, so
`isinstance(m, nn.Conv2d)` style code of the reference - weight init, BN freezing - keeps working) but their
arithmetic is the HIP kernels; `HipSequential` additionally fuses Conv -> BN -> ReLU -> Dropout runs.
"""
import torch
from torch import nn

from . import functional as HF


def _single(v):
    if isinstance(v, (tuple, list)):
        if any(x != v[0] for x in v):
            raise HF.DsrlHipError(f'only square stride/padding/dilation are implemented, got {v}')
        return int(v[0])
    return int(v)


class HipConv2d(nn.Conv2d):
    def forward(self, x, grad_slot=None, in_link=None):
        """grad_slot: functional.GradSlot shared with the other consumers of x (all of them convs or a residual BN), see functional.fork.
        in_link: functional.BNLink of the BatchNorm that produced x, when this conv is its ONLY consumer (the data gradient leaves the BN's backward sums)."""
        if self.groups != 1 or self.padding_mode != 'zeros':
            raise HF.DsrlHipError('HipConv2d: groups=1 and zero padding only')
        k = self.kernel_size
        if self.out_channels == 1 and k == (1, 1) and self.bias is None and self.in_channels % 4 != 0:
            if grad_slot is not None:
                grad_slot.closed = True         # this consumer reports its own gradient: nobody may accumulate into a shared buffer
            # feature transformers, DSRL.py:88-93; x may carry the slot it shares with the fused loss (functional.fused_losses)
            return HF.pointwise_strided(x, self.weight, _single(self.stride), getattr(x, '_dsrl_out_slot', None))
        return HF.conv2d(x, self.weight, self.bias, _single(self.stride), _single(self.padding), _single(self.dilation), grad_slot=grad_slot, in_link=in_link)


class HipBatchNorm2d(nn.BatchNorm2d):
    def forward(self, x):
        return HF.batch_norm_act(x, self)

    def _flush_batches(self):
        n = getattr(self, '_dsrl_batches', 0)
        if n and self.num_batches_tracked is not None:
            self.num_batches_tracked += n
            self._dsrl_batches = 0

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self._flush_batches()
        super()._save_to_state_dict(destination, prefix, keep_vars)


class HipReLU(nn.ReLU):
    def forward(self, x):
        raise HF.DsrlHipError('HipReLU is always fused into the preceding BatchNorm by HipSequential')


class HipDropout(nn.Dropout):
    rng_stream = 0

    def forward(self, x):
        return HF.dropout(x, self.p, self.training, HF.current_seed(), self.rng_stream)


class HipConvTranspose2d(nn.ConvTranspose2d):
    def _apply(self, fn, *args, **kwargs):
        # model.to(memory_format=channels_last) re-lays every 4-D parameter; the k2s2 kernels read this (Cin,Cout,2,2) filter in its logical order,
        # so it is kept contiguous here instead of being copied by a .contiguous() launch in every forward and backward
        r = super()._apply(fn, *args, **kwargs)
        if self.weight.dim() == 4 and not self.weight.is_contiguous():
            with torch.no_grad():
                self.weight.data = self.weight.data.contiguous()
        return r

    def forward(self, x):
        if self.kernel_size != (2, 2) or self.stride != (2, 2) or self.padding != (0, 0) or self.output_padding != (0, 0) or self.groups != 1:
            raise HF.DsrlHipError('HipConvTranspose2d implements kernel_size=2, stride=2, padding=0 (DSRL.py:55-69)')
        # logits_layer (set by the model on the layer whose output goes to the loss): the loss may leave its gradient to this layer's backward
        lg = HF.LogitsGrad() if (getattr(self, 'logits_layer', False) and HF.convt_ce_enabled and torch.is_grad_enabled() and x.requires_grad) else None
        return HF.conv_transpose2d_k2s2(x, self.weight, self.bias, lg)


class HipUpsamplingBilinear2d(nn.UpsamplingBilinear2d):
    def forward(self, x):
        if self.size is not None:
            size = self.size if isinstance(self.size, (tuple, list)) else (self.size, self.size)
        else:
            sf = self.scale_factor if isinstance(self.scale_factor, (tuple, list)) else (self.scale_factor, self.scale_factor)
            size = (int(x.shape[2] * sf[0]), int(x.shape[3] * sf[1]))
        return HF.upsample_bilinear_ac(x, size)


class HipPixelShuffle(nn.PixelShuffle):
    def forward(self, x):
        return HF.pixel_shuffle(x, self.upscale_factor)


class HipAdaptiveAvgPool2d(nn.AdaptiveAvgPool2d):
    def forward(self, x):
        if self.output_size not in (1, (1, 1)):
            raise HF.DsrlHipError('HipAdaptiveAvgPool2d implements output_size=(1,1) (ASPP.py:22)')
        return HF.global_avg_pool(x)


class HipMaxPool2d(nn.MaxPool2d):
    def forward(self, x):
        if (_single(self.kernel_size), _single(self.stride), _single(self.padding), _single(self.dilation)) != (3, 2, 1, 1) or self.ceil_mode:
            raise HF.DsrlHipError('HipMaxPool2d implements kernel 3, stride 2, padding 1 (ResNet101.py:32)')
        return HF.max_pool3x3s2(x)


class HipSequential(nn.Sequential):
    """nn.Sequential that runs Conv2d -> BatchNorm2d -> ReLU -> Dropout as conv + one fused BN/activation pass."""

    def forward(self, x, residual=None, grad_slot=None, out_link=None):
        """grad_slot goes to the first module when that is a HipConv2d (the consumer of x).  BatchNorm-backward links (functional.BNLink, round 5): the
        output of a fused conv -> BN -> ReLU (-> Dropout) group that is followed by another conv -> BN group feeds ONLY that conv, so its data gradient
        can leave the BN's two per-channel sums (cat_conv: DSRL.py:34-49); `out_link` does the same for the last group when the caller knows the single
        consumer of this Sequential's output (DSRL.forward_head: cls_conv)."""
        mods = list(self)
        i, n = 0, len(mods)
        link_in = None              # link of the tensor x currently is, if a BN group just produced it
        if grad_slot is not None and not (n and isinstance(mods[0], HipConv2d)):
            grad_slot.closed = True             # nobody here can honour the slot
        seed = HF.current_seed()
        while i < n:
            m = mods[i]
            if isinstance(m, nn.BatchNorm2d):
                relu = i + 1 < n and isinstance(mods[i + 1], nn.ReLU)
                j = i + (2 if relu else 1)
                p, stream = 0.0, 0
                if relu and j < n and isinstance(mods[j], nn.Dropout):
                    if mods[j].training and mods[j].p > 0:
                        p, stream = mods[j].p, getattr(mods[j], 'rng_stream', 0)
                    j += 1
                last = j >= n
                x = HF.batch_norm_act(x, m, relu=relu, drop_p=p, seed=seed, rng_stream=stream, residual=residual if last else None)
                i = j
            elif isinstance(m, nn.ReLU):
                raise HF.DsrlHipError('HipSequential: ReLU must follow a BatchNorm2d')
            elif (isinstance(m, HipConv2d) and i + 1 < n and isinstance(mods[i + 1], nn.BatchNorm2d) and m.groups == 1 and m.padding_mode == 'zeros'
                  and not (m.out_channels == 1 and m.kernel_size == (1, 1) and m.bias is None and m.in_channels % 4 != 0)):
                # conv -> BN (-> ReLU -> Dropout): the conv epilogue provides the BN statistics when it can (HF.conv2d_bn_act)
                bn = mods[i + 1]
                relu = i + 2 < n and isinstance(mods[i + 2], nn.ReLU)
                j = i + (3 if relu else 2)
                p, stream = 0.0, 0
                if relu and j < n and isinstance(mods[j], nn.Dropout):
                    if mods[j].training and mods[j].p > 0:
                        p, stream = mods[j].p, getattr(mods[j], 'rng_stream', 0)
                    j += 1
                # does the output feed exactly one fused conv -> BN group of this Sequential (or, for the last group, the caller's single consumer)?
                nxt = (j + 1 < n and isinstance(mods[j], HipConv2d) and isinstance(mods[j + 1], nn.BatchNorm2d) and mods[j].groups == 1 and mods[j].in_channels % 32 == 0)
                lout = None
                if HF.bn_bwd_stats_enabled and torch.is_grad_enabled() and residual is None:
                    lout = HF.BNLink() if nxt else (out_link if j >= n else None)
                x = HF.conv2d_bn_act(x, m.weight, m.bias, _single(m.stride), _single(m.padding), _single(m.dilation), bn, relu=relu, drop_p=p, seed=seed,
                                     rng_stream=stream, residual=residual if j >= n else None, grad_slot=grad_slot if i == 0 else None,
                                     in_link=link_in, out_link=lout)
                link_in = lout if nxt else None
                i = j
            else:
                x = m(x, grad_slot=grad_slot) if (i == 0 and grad_slot is not None and isinstance(m, HipConv2d)) else m(x)
                i += 1
        return x
