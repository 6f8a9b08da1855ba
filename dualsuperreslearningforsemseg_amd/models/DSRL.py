"""Dual Super-Resolution Learning model on the MI355X kernels - the surface of the reference's models/DSRL.py:11-186:
`DSRL(stage, dataset_settings, init_weights=True, BatchNorm2d=...)`, the same sub-module attribute names and therefore the
same 702 state_dict keys, `.forward(x) -> (SSSR, SISR, SSSR_ft, SISR_ft)`.  Tensors are logical NCHW in
torch.channels_last memory; call `model.to(device).to(memory_format=torch.channels_last)` to store conv weights in
the [K][R][S][C] layout the kernels read without a per-step re-layout.
"""
import torch as t

from .. import consts
from .. import functional as HF
from ..nn_modules import (HipBatchNorm2d, HipConv2d, HipConvTranspose2d, HipDropout, HipPixelShuffle, HipReLU, HipSequential,
                          HipUpsamplingBilinear2d)
from .BaseModel import BaseModel
from .modules.ASPP import ASPP
from .modules.backbone import ResNet101

# Philox stream id of each Dropout module (shared with oracle.dsrl_oracle.DROPOUT_STREAMS)
_DROPOUT_STREAMS = {('cat_conv', 3): 1, ('cat_conv', 7): 2, ('upsample16_pred', 1): 3, ('upsample16_pred', 5): 4}


class DSRL(BaseModel):

    @staticmethod
    def _define_feature_extractor(in_channels: int, out_channels1: int, out_channels2: int):
        return t.nn.ModuleDict({
            'backbone': ResNet101(replace_stride_with_dilation=[False, False, True]),                       # DSRL.py:17
            'aspp': ASPP(in_channels=in_channels, out_channels=out_channels1, rate=1),                       # DSRL.py:18
            'shortcut_conv': HipSequential(HipConv2d(out_channels1, out_channels2, kernel_size=1, padding=0, bias=False),
                                           HipBatchNorm2d(num_features=out_channels2), HipReLU())})        # DSRL.py:19-25

    @staticmethod
    def _define_SSSR_decoder(in_channels1: int, in_channels2: int, mid_channels: int, out_channels: int):
        mods = t.nn.ModuleDict({
            'cat_conv': HipSequential(HipConv2d(in_channels1 + in_channels2, mid_channels, kernel_size=3, padding=1, bias=False),
                                      HipBatchNorm2d(num_features=mid_channels), HipReLU(), HipDropout(p=0.2),
                                      HipConv2d(mid_channels, mid_channels, kernel_size=3, padding=1, bias=False),
                                      HipBatchNorm2d(num_features=mid_channels), HipReLU(), HipDropout(p=0.2)),     # DSRL.py:34-49
            'cls_conv': HipConv2d(mid_channels, out_channels, kernel_size=1, bias=True),                             # DSRL.py:50
            'upsample16_pred': HipSequential(HipUpsamplingBilinear2d(scale_factor=2.0), HipDropout(p=0.2),
                                             HipConvTranspose2d(out_channels, out_channels, kernel_size=2, stride=2, padding=0, bias=False),
                                             HipBatchNorm2d(num_features=out_channels), HipReLU(), HipDropout(p=0.2),
                                             HipConvTranspose2d(out_channels, out_channels, kernel_size=2, stride=2, padding=0, bias=True))})  # DSRL.py:53-69
        for (name, idx), stream in _DROPOUT_STREAMS.items():
            mods[name][idx].rng_stream = stream
        mods['upsample16_pred'][6].logits_layer = True        # its output is SSSR_output: the CE gradient can be formed inside its backward (HF.LogitsGrad)
        return mods

    @staticmethod
    def _define_SISR_decoder(in_channels: int, out_channels: int, upscale_factor: int):
        assert type(upscale_factor) == int, "BUG CHECK: 'upscale_factor' must be an integer type."
        return HipSequential(HipConv2d(in_channels, out_channels * (upscale_factor ** 2), kernel_size=3, stride=1, padding=1, bias=True),
                             HipPixelShuffle(upscale_factor=upscale_factor))                                                # DSRL.py:78-84

    @staticmethod
    def _define_feature_transformer(in_channels: int, out_channels: int):
        return HipSequential(HipConv2d(in_channels, out_channels, kernel_size=1, stride=8, padding=0, bias=False),
                             HipBatchNorm2d(num_features=out_channels), HipReLU())                                          # DSRL.py:88-95

    def __init__(self, stage, dataset_settings, init_weights=True, BatchNorm2d=HipBatchNorm2d):
        assert stage in [1, 2, 3], "BUG CHECK: Unsupported stage {0} specified in DSRL.__init__().".format(stage)
        super().__init__()
        self.stage = stage
        self.feature_extractor = DSRL._define_feature_extractor(in_channels=2048, out_channels1=256, out_channels2=48)
        self.SSSR_decoder = DSRL._define_SSSR_decoder(in_channels1=256, in_channels2=48, mid_channels=256,
                                                      out_channels=dataset_settings.NUM_CLASSES)
        if init_weights:
            self._init_weights(BatchNorm2d, self.feature_extractor['shortcut_conv'], self.SSSR_decoder)
        if self.stage > 1:
            self.SISR_decoder = DSRL._define_SISR_decoder(in_channels=(256 + 48), out_channels=consts.NUM_RGB_CHANNELS, upscale_factor=8)
            if init_weights:
                self._init_weights(BatchNorm2d, self.SISR_decoder)
            if self.stage > 2:
                self.SSSR_feature_transformer = DSRL._define_feature_transformer(in_channels=dataset_settings.NUM_CLASSES, out_channels=1)
                self.SISR_feature_transformer = DSRL._define_feature_transformer(in_channels=consts.NUM_RGB_CHANNELS, out_channels=1)
                if init_weights:
                    self._init_weights(BatchNorm2d, self.SSSR_feature_transformer, self.SISR_feature_transformer)

    @t.no_grad()
    def _init_weights(self, BatchNorm2d, *modules):
        # DSRL.py:143-151 (conv biases keep torch's default init)
        for module in modules:
            for m in module.modules():
                if isinstance(m, (t.nn.Conv2d, t.nn.ConvTranspose2d)):
                    t.nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
                elif isinstance(m, BatchNorm2d):
                    m.weight.fill_(1.0)
                    m.bias.zero_()

    def initialize_with_pretrained_weights(self, weights_dir, map_location=t.device('cpu')):
        self.feature_extractor['backbone'].initialize_with_pretrained_weights(weights_dir, map_location)

    def forward_head(self, backbone_features: t.Tensor, lowlevel_features: t.Tensor):
        """DSRL.py:162-184 on given backbone outputs (the part SURVEY.md section 8 scopes; used directly by the parity tests)."""
        fe = self.feature_extractor
        aspp_features = fe['aspp'](backbone_features)
        h, w = aspp_features.shape[-2:]
        aspp_features = HF.upsample_bilinear_ac(aspp_features, (4 * h, 4 * w))                 # DSRL.py:163
        oslot = getattr(lowlevel_features, '_dsrl_outer_slot', None)                          # shared with layer2's first block (ResNet101.forward)
        lowlevel_features = (fe['shortcut_conv'](lowlevel_features, grad_slot=oslot) if (oslot is not None and isinstance(fe['shortcut_conv'], HipSequential))
                             else fe['shortcut_conv'](lowlevel_features))                     # DSRL.py:164
        cat_features = HF.cat_channels([aspp_features, lowlevel_features])                    # DSRL.py:165
        # cat_features feeds cat_conv.0 and (stage > 1) the SISR conv: both data gradients accumulate in one buffer (HF.GradSlot)
        slot = None
        if (HF.grad_slots_enabled and self.stage > 1 and cat_features.requires_grad and t.is_grad_enabled()
                and isinstance(self.SSSR_decoder['cat_conv'], HipSequential) and isinstance(self.SISR_decoder, HipSequential)):
            cat_features, slot = HF.fork(cat_features), HF.GradSlot()
        # the output of cat_conv (BN -> ReLU -> Dropout, DSRL.py:44-49) feeds cls_conv only: its data gradient leaves that BatchNorm's backward sums
        link = HF.BNLink() if (HF.bn_bwd_stats_enabled and t.is_grad_enabled() and isinstance(self.SSSR_decoder['cat_conv'], HipSequential)
                               and isinstance(self.SSSR_decoder['cls_conv'], HipConv2d)) else None
        kw = {} if link is None else {'out_link': link}
        SSSR_output = (self.SSSR_decoder['cat_conv'](cat_features, grad_slot=slot, **kw) if slot is not None
                       else self.SSSR_decoder['cat_conv'](cat_features, **kw))                # DSRL.py:168
        SSSR_output = (self.SSSR_decoder['cls_conv'](SSSR_output, in_link=link) if (link is not None and link.valid)
                       else self.SSSR_decoder['cls_conv'](SSSR_output))                       # DSRL.py:169
        SSSR_output = self.SSSR_decoder['upsample16_pred'](SSSR_output)                       # DSRL.py:170
        # DSRL.py:172-174: unused outputs are CPU zeros(1) whatever the model device
        SISR_output = t.zeros(1, requires_grad=False)
        SSSR_transform_output = t.zeros(1, requires_grad=False)
        SISR_transform_output = t.zeros(1, requires_grad=False)
        if self.stage > 1:
            SISR_output = self.SISR_decoder(cat_features, grad_slot=slot) if slot is not None else self.SISR_decoder(cat_features)   # DSRL.py:177
            if self.stage > 2:
                if HF.grad_slots_enabled and t.is_grad_enabled() and SSSR_output.requires_grad:
                    # each output feeds its feature transformer and the loss: with functional.fused_losses the loss publishes the dense
                    # gradient and the stride-8 transformer adds its sparse one into it (no 320 MB mostly-zero tensor + add pass)
                    SSSR_output._dsrl_out_slot, SISR_output._dsrl_out_slot = HF.GradSlot(), HF.GradSlot()
                SSSR_transform_output = self.SSSR_feature_transformer(SSSR_output)            # DSRL.py:181
                SISR_transform_output = self.SISR_feature_transformer(SISR_output)            # DSRL.py:184
        return SSSR_output, SISR_output, SSSR_transform_output, SISR_transform_output

    def forward(self, x: t.Tensor):
        with t.autograd.profiler.record_function(DSRL.forward.__qualname__):                  # DSRL.py:159
            HF.begin_forward(self.training)
            backbone_features, lowlevel_features = self.feature_extractor['backbone'](x)      # DSRL.py:161
            return self.forward_head(backbone_features, lowlevel_features)
