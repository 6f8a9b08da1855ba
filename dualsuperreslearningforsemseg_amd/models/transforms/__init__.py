"""Device-side tail of the reference's training input pipeline (models/transforms/): everything after the random PIL
augmentations is deterministic - label-id remap (JointImageAndLabelTensor.py:9-16), ToTensor + Normalize (JointNormalize.py:11)
and the dual-scale resize (JointScaledImage.py:27-32) - and runs as one HIP call on decoded uint8 crops."""
import ctypes

import torch

from ... import functional as HF
from ..._lib import call


class DeviceBatchPreparation:
    def __init__(self, label_mapping_dict, mean, std, model_input_size, ignore_label=255):
        lut = torch.full((256,), ignore_label, dtype=torch.uint8)
        for k, v in label_mapping_dict.items():
            if 0 <= k < 256:
                lut[k] = v
        self.lut_host = lut
        self._lut = {}
        self.mean = (ctypes.c_float * 3)(*mean)
        self.std = (ctypes.c_float * 3)(*std)
        self.size = tuple(model_input_size)

    def __call__(self, rgb_u8, labels_u8=None):
        """rgb_u8 (N,Hs,Ws,3) uint8, labels_u8 (N,Hs,Ws) uint8 raw label ids, both on the device.
        Returns ((input_image, input_org), (target, labels_u8)) like the reference's loader (JointScaledImage.py:31-32):
        input_image (N,3,H,W) and input_org (N,3,2H,2W) are channels_last views (input_image carries a zero 4th channel for the stem)."""
        HF._need_gpu(rgb_u8, labels_u8)
        N, Hs, Ws, _ = rgb_u8.shape
        H, W = self.size
        dev = rgb_u8.device
        lut = self._lut.get(dev)
        if lut is None:
            lut = self._lut[dev] = self.lut_host.to(dev)
        img_in = torch.empty((N, H, W, 4), device=dev, dtype=torch.float32)
        img_org = torch.empty((N, 2 * H, 2 * W, 3), device=dev, dtype=torch.float32)
        target = torch.empty((N, 2 * H, 2 * W), device=dev, dtype=torch.uint8) if labels_u8 is not None else None
        call('dsrl_prepare_batch', rgb_u8.contiguous().data_ptr(), None if labels_u8 is None else labels_u8.contiguous().data_ptr(), lut.data_ptr(),
             self.mean, self.std, img_in.data_ptr(), img_org.data_ptr(), None if target is None else target.data_ptr(), N, Hs, Ws, H, W, HF._stream())
        return (img_in.permute(0, 3, 1, 2)[:, :3], img_org.permute(0, 3, 1, 2)), (target, labels_u8)
