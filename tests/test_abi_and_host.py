"""CPU-only checks: the C-ABI library loads and exports every symbol include/dsrl_hip.h declares (no compute calls), the
ctypes prototype table covers the header, the module surface has the reference's state_dict keys, and the host logic
(pixel-major stride detection, polynomial LR, loud failure without a GPU)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'dsrl_hip.h')


def header_symbols():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(dsrl_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_header_symbol():
    from dualsuperreslearningforsemseg_amd import _lib
    assert os.path.isfile(_lib.LIB_PATH), 'run __graft_entry__.build() first'
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 50
    for s in syms:
        assert hasattr(lib, s), f'{s} declared in include/dsrl_hip.h but not exported'
    assert sorted(_lib.PROTOTYPES) == syms, set(_lib.PROTOTYPES) ^ set(syms)
    loaded = _lib.load()
    assert loaded.dsrl_version() == 1
    # pure host helpers may be called without a GPU
    assert loaded.dsrl_conv2d_inbounds_macs(1, 16, 32, 2048, 256, 3, 3, 1, 18, 18) == 503316480        # d18 -> 503.3 M
    assert loaded.dsrl_conv2d_inbounds_macs(1, 16, 32, 2048, 256, 3, 3, 1, 6, 6) == 1585446912
    assert loaded.dsrl_conv2d_inbounds_macs(1, 64, 128, 304, 256, 3, 3, 1, 1, 1) == 190 * 382 * 304 * 256  # 5,648.5 M
    assert loaded.dsrl_conv2d_fwd_workspace_bytes(8, 64, 128, 304, 256, 3, 3, 1, 1, 1) == 2048     # no split-K slabs: only the two amax records of the f16x3 arithmetic
    assert loaded.dsrl_conv2d_fwd_workspace_bytes(8, 16, 32, 2048, 256, 3, 3, 1, 6, 6) > 0            # split-K slabs


def test_head_macs_match_survey():
    """In-bounds MACs of the decoder-head conv stack at 256x512 = 18,450.2 M."""
    from dualsuperreslearningforsemseg_amd import _lib
    f = _lib.load().dsrl_conv2d_inbounds_macs
    convs = [(16, 32, 2048, 256, 1, 0, 1), (16, 32, 2048, 256, 3, 6, 6), (16, 32, 2048, 256, 3, 12, 12), (16, 32, 2048, 256, 3, 18, 18),
             (1, 1, 2048, 256, 1, 0, 1), (16, 32, 1280, 256, 1, 0, 1), (64, 128, 256, 48, 1, 0, 1), (64, 128, 304, 256, 3, 1, 1),
             (64, 128, 256, 256, 3, 1, 1), (64, 128, 256, 19, 1, 0, 1), (64, 128, 304, 192, 3, 1, 1)]
    total = sum(f(1, h, w, c, k, r, r, 1, p, d) for h, w, c, k, r, p, d in convs)
    total += 128 * 256 * 19 * 19 * 4 + 256 * 512 * 19 * 19 * 4 + 64 * 128 * 19 + 64 * 128 * 3       # convT x2 + feature transformers
    assert abs(total - 18450.2e6) < 0.05e6, total


def test_cpu_tensors_fail_loudly():
    import dualsuperreslearningforsemseg_amd as D
    from dualsuperreslearningforsemseg_amd import functional as HF
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        HF.conv2d(torch.zeros(1, 4, 4, 4), torch.zeros(4, 4, 1, 1))
    with pytest.raises(RuntimeError):
        D.FALoss()(torch.zeros(2, 1, 64, 128), torch.zeros(2, 1, 64, 128))
    with pytest.raises(AssertionError):                       # the reference's BUG CHECK asserts come first (FALoss.py:19-20)
        D.FALoss()(torch.zeros(2, 1, 64), torch.zeros(2, 1, 64))


def test_state_dict_surface():
    import dualsuperreslearningforsemseg_amd as D
    from dualsuperreslearningforsemseg_amd.datasets.Cityscapes import settings as cs
    counts = {}
    for stage in (1, 2, 3):
        m = D.DSRL(stage, cs)
        counts[stage] = sum(p.numel() for p in m.parameters())
    assert counts == {1: 59346740, 2: 59872244, 3: 59872270}
    sd = m.state_dict()
    assert len(sd) == 702
    head = [k for k in sd if 'backbone' not in k]
    assert len(head) == 78
    for k in ('feature_extractor.aspp.branches.5.1.running_var', 'feature_extractor.shortcut_conv.0.weight', 'SSSR_decoder.cat_conv.4.weight',
              'SSSR_decoder.cls_conv.bias', 'SSSR_decoder.upsample16_pred.2.weight', 'SSSR_decoder.upsample16_pred.6.bias', 'SISR_decoder.0.bias',
              'SSSR_feature_transformer.1.num_batches_tracked', 'feature_extractor.backbone.layer3.22.conv3.weight',
              'feature_extractor.backbone.layer2.0.downsample.1.running_mean', 'feature_extractor.backbone.conv1.weight'):
        assert k in sd, k
    assert tuple(sd['SSSR_decoder.upsample16_pred.2.weight'].shape) == (19, 19, 2, 2)
    assert tuple(sd['SISR_decoder.0.weight'].shape) == (192, 304, 3, 3)
    assert m.stage == 3 and isinstance(m.feature_extractor['aspp'], D.ASPP)
    # torch-class compatibility the reference code relies on (weight init / BN freezing by isinstance)
    assert isinstance(m.SSSR_decoder['cat_conv'][0], torch.nn.Conv2d) and isinstance(m.SSSR_decoder['cat_conv'][1], torch.nn.BatchNorm2d)


def test_pixel_major_stride_detection():
    from dualsuperreslearningforsemseg_amd.functional import _ld_of
    x = torch.zeros(2, 8, 4, 6).contiguous(memory_format=torch.channels_last)
    assert _ld_of(x) == 8 and _ld_of(x[:, 2:6]) == 8 and _ld_of(torch.zeros(2, 8, 4, 6)) is None
    assert _ld_of(torch.zeros(2, 1, 4, 6)) == 1 and _ld_of(torch.zeros(3, 5, 1, 1)) == 5


def test_polynomial_lr_matches_reference_formula():
    from dualsuperreslearningforsemseg_amd.command_handlers.train_or_resume import polynomial_lr
    assert polynomial_lr(0.006, 0.0005, 0, 250, 0.9) == 0.006
    assert abs(polynomial_lr(0.006, 0.0005, 125, 250, 0.9) - ((0.006 - 0.0005) * 0.5 ** 0.9 + 0.0005)) < 1e-12
    assert abs(polynomial_lr(0.006, 0.0005, 250, 250, 0.9) - 0.0005) < 1e-12


def test_train_or_resume_rejects_cpu_and_amp():
    from dualsuperreslearningforsemseg_amd.command_handlers.train_or_resume import train_or_resume
    kw = dict(is_resuming_training=False, distributed=None, disable_cudnn_benchmark=False, num_workers=0, dataset={}, val_interval=1,
              checkpoint_interval=1, checkpoint_history=1, init_weights=None, batch_size=2, epochs=1, learning_rate=0.01, end_learning_rate=0.001,
              momentum=0.9, weights_decay=5e-4, poly_power=0.9, stage=3, w1=0.1, w2=1.0, freeze_batch_norm=False, experiment_id='', description='',
              early_stopping=False)
    with pytest.raises(RuntimeError):
        train_or_resume(device='cpu', mixed_precision=None, **kw)
    with pytest.raises(RuntimeError):
        train_or_resume(device='gpu', mixed_precision='fp8', **kw)          # apex opt levels O0..O3 select the conv arithmetic; anything else is refused


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No .so -> DsrlHipError with the build hint; nothing falls back to another code path."""
    from dualsuperreslearningforsemseg_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'libdsrl_hip.so'))
    with pytest.raises(_lib.DsrlHipError, match='no fallback'):
        _lib.load()


def test_product_never_imports_the_oracle():
    import subprocess, sys
    code = "import sys; import dualsuperreslearningforsemseg_amd, dualsuperreslearningforsemseg_amd.command_handlers.train_or_resume; " \
           "import dualsuperreslearningforsemseg_amd.metrices, dualsuperreslearningforsemseg_amd.models.transforms; " \
           "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'oracle imported'"
    subprocess.run([sys.executable, '-c', code], check=True, cwd=ROOT)


def test_conv_precision_api_roundtrip():
    """functional.set_conv_precision / get_conv_precision drive dsrl_conv_precision (include/dsrl_hip.h); no GPU needed."""
    from dualsuperreslearningforsemseg_amd import functional as HF
    HF.set_conv_precision(None)
    default = HF.get_conv_precision()
    assert default == {'0': 'fp32', '1': 'bf16x3', '2': 'bf16x6', '3': 'mixed', '4': 'f16x3', '5': 'f16x1'}[os.environ.get('DSRL_CONV_PRECISION', '4')]
    for mode in ('fp32', 'bf16x3', 'bf16x6', 'mixed', 'f16x3', 'f16x1'):
        HF.set_conv_precision(mode)
        assert HF.get_conv_precision() == mode
    with pytest.raises((KeyError, ValueError)):
        HF.set_conv_precision('fp8')
    HF.set_conv_precision(None)
    assert HF.get_conv_precision() == default


def test_bn_fused_block_budget_roundtrip():
    """functional.set_bn_fused_max_blocks drives dsrl_bn_fused_max_blocks (0 = off, 128, 256, None = environment); no GPU needed."""
    from dualsuperreslearningforsemseg_amd import functional as HF
    first = HF.set_bn_fused_max_blocks(128)
    assert HF.set_bn_fused_max_blocks(0) == 128
    assert HF.set_bn_fused_max_blocks(256) == 0
    assert HF.set_bn_fused_max_blocks(None) == 256
    assert HF.set_bn_fused_max_blocks(first if first >= 0 else None) == -1


def test_concat_and_residual_sum_queries_host_side():
    """Host-only halves of two round-5 entry points (no GPU needed): dsrl_cat_channels_supported checks source count, widths / strides (multiples of 4),
    alignment and the 32-bit float4 index range; dsrl_bn_bwd_from_stats_res_parts reports the row blocks of residual-branch sums the backward launch of a
    [P][C] tensor writes (0 for shapes that launch cannot take) and never more than the partials buffer is sized for."""
    import ctypes
    from dualsuperreslearningforsemseg_amd import functional as HF

    def ok(ptrs, lds, cs, dst, ld_dst, P):
        n = len(ptrs)
        return HF.query('dsrl_cat_channels_supported', (ctypes.c_void_p * n)(*ptrs), (ctypes.c_int32 * n)(*lds), (ctypes.c_int32 * n)(*cs), n, dst, ld_dst, P)

    base = 1 << 20
    assert ok([base, base + 4096], [256, 48], [256, 48], base + 8192, 304, 65536) == 1            # DSRL.py:165
    assert ok([base + 1024 * i for i in range(5)], [256] * 5, [256] * 5, base, 1280, 4096) == 1    # ASPP.py:44
    assert ok([base, base + 4096], [19, 48], [19, 48], base + 8192, 67, 65536) == 0               # widths not multiples of 4: per-source copies
    assert ok([base + 4, base + 4096], [256, 48], [256, 48], base + 8192, 304, 65536) == 0        # unaligned source
    assert ok([base] * 9, [4] * 9, [4] * 9, base, 36, 16) == 0                                    # more than eight sources
    assert ok([base, base + 4096], [256, 48], [256, 48], base + 8192, 304, 1 << 26) == 0          # 2^26 pixels x 76 float4: beyond the 32-bit index
    for P, C, parts in ((65536, 256, 32), (16384, 512, 128), (4096, 1024, 64), (4096, 2048, 32), (1000, 64, 1), (65536, 64, 1024)):
        n = HF.query('dsrl_bn_bwd_from_stats_res_parts', P, C, parts)
        assert 0 < n <= 1024 and HF.query('dsrl_bn_stats_floats', 2, n, C) >= 2 * n * C, (P, C, parts, n)
    assert HF.query('dsrl_bn_bwd_from_stats_res_parts', 4096, 48, 32) == 0                         # C % 32 != 0: no from-statistics launch
    assert HF.query('dsrl_bn_bwd_from_stats_res_parts', 4096, 256, 0) == 0


@pytest.mark.parametrize('unit,least', [('conv_planes', 100), ('convt_dma', 10)])
def test_lds_dma_inline_asm_is_the_only_m0_user(tmp_path, unit, least):
    """lds_dma.h sets M0 inside inline asm without declaring it (hipcc refuses "m0" as a clobber: reserved register).  That is sound only while
    nothing else in the code objects that include it touches M0: disassemble them and require every instruction that names m0 to be one of those
    s_mov_b32."""
    import shutil, subprocess
    objdump = '/opt/rocm/lib/llvm/bin/llvm-objdump'
    obj = os.path.join(ROOT, 'dualsuperreslearningforsemseg_amd', 'csrc', unit + '.o')
    if not (os.path.isfile(objdump) and os.path.isfile(obj)):
        pytest.skip('needs llvm-objdump and the in-tree object file')
    local = tmp_path / (unit + '.o')
    shutil.copy(obj, local)
    subprocess.run([objdump, '--offloading', str(local)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp_path)
    dev = [p for p in os.listdir(tmp_path) if 'gfx950' in p]
    assert len(dev) == 1, os.listdir(tmp_path)
    asm = subprocess.run([objdump, '-d', str(tmp_path / dev[0])], check=True, capture_output=True, text=True).stdout
    users = [ln.split('//')[0].split() for ln in asm.splitlines() if re.search(r'\bm0\b', ln.split('//')[0])]
    assert len(users) > least                                 # the DMA pieces are there
    odd = [u for u in users if not (u[0] == 's_mov_b32' and u[1] == 'm0,' and re.fullmatch(r's\d+', u[2]))]
    assert not odd, odd[:5]
    # and no other translation unit of the library includes the helper
    for src in os.listdir(os.path.dirname(obj)):
        if src.endswith('.hip') and src[:-4] not in ('conv_planes', 'convt_dma'):
            assert 'lds_dma.h' not in open(os.path.join(os.path.dirname(obj), src)).read(), src


def test_no_new_register_spills():
    """Scratch (spilled registers, private arrays) in a kernel of the library is a decision, not an accident: every kernel of the in-tree objects that needs
    scratch must be on this list, with where it can still run (round 5 removed the 256x256 data-gradient and the four-group
    weight-gradient builds, which the planner never picked)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import kernel_resources as KR
    if not os.path.isfile(os.path.join(KR.LLVM, 'llvm-objdump')):
        pytest.skip('needs llvm-objdump')
    allowed = [
        r'conv_igemm_split_kernel<4, 2, 2, 4, false, [12], 1, [12], false, false>',   # 256x256 forward, register-staged: only when a decoder-size conv arrives without planes (f16x1 on-the-fly filter build: 4 registers)
        r'conv_igemm_f32_kernel<2, 2, 2, 2, (true|false), 4, false>',                  # exact-fp32 mode, <= 128-register build that keeps 4 blocks per CU (a measured win in that mode)
        r'convt2x2_bwd_fused_kernel<(19, 19|8, 8)>',                                   # VALU fallback of the ConvTranspose backward (W % 4 != 0 or DSRL_CONVT_MFMA=0)
    ]
    csrc = os.path.join(ROOT, 'dualsuperreslearningforsemseg_amd', 'csrc')
    objs = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith('.o')]
    if not objs:
        pytest.skip('no in-tree objects (run __graft_entry__.build() first)')
    bad = []
    for o in objs:
        ks = KR.kernels(o)
        names = KR.demangle([k['name'] for k in ks])
        for k, n in zip(ks, names):
            if (k.get('scratch', 0) or k.get('vgpr_spill', 0)) and not any(re.search(a, n) for a in allowed):
                bad.append((os.path.basename(o), n[:120], k.get('scratch', 0), k.get('vgpr_spill', 0)))
    assert not bad, bad
