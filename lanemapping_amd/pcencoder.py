"""BEV feature encoder behind the reference's PCENCODER registry name ``PostProjector2``.

Drop-in for baseline/models/pcencoder/postprojector.py: same constructor kwargs (:59-65), same
``forward(sample) -> (fea_downsample, fea_up, out_binary_seg, out_endp_seg)`` contract (:79-82), same
state-dict key layout (incl. the never-executed ``fpn.model_buttomup.*`` ResNet-34) so a
reference checkpoint loads with ``strict=True``.  The arithmetic runs in liblanemap_hip.so:

  stem 7x7+BN+ReLU, max-pool                      -> lm_stem_conv7x7_bn_relu, lm_maxpool3x3s2_nhwc
  BasicBlocks, lateral/top/smooth/semantic convs  -> lm_conv2d_nhwc_mfma_f32 (BN / bias / residual / ReLU epilogue)
  GroupNorm(C,C)+ReLU+bilinear(+sum of branches)  -> lm_conv2d_nhwc_mfma_f32_gnstats + lm_gn_finalize (statistics out of the
                                                     conv epilogue) / lm_gn_stats, then lm_gn_relu_upsample
  1x1 heads (128->8, 8->3, 128->1)                -> lm_conv2d_nhwc_small
  final 4x bilinear to the tile resolution        -> lm_upsample_bilinear_to_chw
"""
import os

import torch
import torch.nn as nn

from . import ops, trace
from .registry import PCENCODER
from .packing import PackedModule

_LAYERS = {'resnet18': [2, 2, 2, 2], 'resnet34': [3, 4, 6, 3]}
MERGE_BRANCH_CONVS = os.environ.get('LANEMAP_MERGE_BRANCH_CONVS', '1') != '0'   # conv_b of both semantic branches on p2 / p3 as one launch
# Winograd F(4x4,3x3) (csrc/conv_wino44.hip): every 3x3 / stride-1 convolution with >= WINO_F44_MIN_CIN input channels that the kernel
# supports (Cin % 16 == 0, tile rows wide enough: lm_winograd44_supported) - 36 products per 4x4 output block instead of 144, exact fp32
# MFMA, no transformed tensor in HBM.  Everything else, and everything under LANEMAP_WINO_F44=0, runs the direct implicit-GEMM kernel
# (csrc/conv_mfma.hip).  The F(2x2,3x3) family of rounds 1-3 was removed in round 5 (history: profiles/README.md).
WINO_F44 = os.environ.get('LANEMAP_WINO_F44', '1') != '0'
WINO_F44_MIN_CIN = int(os.environ.get('LANEMAP_WINO_F44_MIN_CIN', '64'))
# SECOND LINE, never the default: LANEMAP_WINO_SPLIT=1 runs the same F(4x4) kernel with its Winograd-domain products on the fp16 matrix
# pipe (every fp32 operand in two fp16 terms, three products, fp32 accumulation: csrc/conv_wino44.hip, w44_split2).  Not bit-identical to
# the exact path; bench.py reports it as `second_line`, the headline stays exact fp32.
WINO_SPLIT = os.environ.get('LANEMAP_WINO_SPLIT', '0') != '0'


def _frag44(w):
    if WINO_SPLIT:
        return ops.pack_wino44_fragments_split(ops.pack_wino44(w))
    return ops.pack_wino44_fragments(ops.pack_wino44(w))


def _f44_ok(conv):
    return WINO_F44 and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.in_channels >= WINO_F44_MIN_CIN and conv.in_channels % 16 == 0


class _ResBlock(nn.Module):
    """Parameter container of one BasicBlock (reference :299-338)."""

    def __init__(self, cin, cout, stride, dilation, with_down):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, dilation, dilation, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, dilation, dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout)) if with_down else None
        self.stride, self.dilation = stride, dilation


def _stage(state, planes, blocks, stride, dilate):
    """Reference `_make_layer` (:517-539): `state` = [inplanes, dilation]."""
    prev_dil = state[1]
    if dilate:
        state[1] *= stride
        stride = 1
    layers = [_ResBlock(state[0], planes, stride, prev_dil, stride != 1 or state[0] != planes)]
    state[0] = planes
    for _ in range(1, blocks):
        layers.append(_ResBlock(planes, planes, 1, state[1], False))
    return nn.Sequential(*layers)


class _BottomUp(nn.Module):
    """The reference builds a second ResNet (`model_buttomup`, :436-438) that forward never calls;
    kept only so checkpoints load strictly."""

    def __init__(self, layers, dil, in_channels):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        st = [64, 1]
        self.layer1 = _stage(st, in_channels[0], layers[0], 1, False)
        self.layer2 = _stage(st, in_channels[1], layers[1], 2, dil[0])
        if in_channels[2] > 0:
            self.layer3 = _stage(st, in_channels[2], layers[2], 2, dil[1])
        if in_channels[3] > 0:
            self.layer4 = _stage(st, in_channels[3], layers[3], 2, dil[2])


class FPNEncoder(PackedModule):
    def __init__(self, resnet='resnet34', pretrained=False, replace_stride_with_dilation=(False, True, False),
                 out_conv=True, in_channels=(64, 128, 256, -1), cfg=None):
        super().__init__()
        if resnet not in _LAYERS:
            raise NotImplementedError(f'{resnet}: only BasicBlock ResNets are on the hot path')
        if in_channels[2] <= 0 or in_channels[3] > 0:
            raise NotImplementedError('hot path covers in_channels=[c1,c2,c3,-1] (all BASELINE configs)')
        dil = list(replace_stride_with_dilation)
        if len(dil) != 3:
            raise ValueError(f'replace_stride_with_dilation should be None or a 3-element tuple, got {dil}')
        layers = _LAYERS[resnet]
        self.cfg = cfg
        self.in_channels = list(in_channels)
        self.model_buttomup = _BottomUp(layers, dil, in_channels)
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        st = [64, 1]
        self.layer1 = _stage(st, in_channels[0], layers[0], 1, False)
        self.layer2 = _stage(st, in_channels[1], layers[1], 2, dil[0])
        self.layer3 = _stage(st, in_channels[2], layers[2], 2, dil[1])
        c = st[0]                                   # 256
        self.out = nn.Conv2d(c, cfg.featuremap_out_channel, 1, bias=False) if out_conv else None
        self.toplayer = nn.Conv2d(c, 256, 1)
        self.smooth1 = nn.Conv2d(c, c, 3, 1, 1)
        self.smooth2 = nn.Conv2d(c, c, 3, 1, 1)
        self.smooth3 = nn.Conv2d(c, c, 3, 1, 1)
        self.latlayer1 = nn.Conv2d(in_channels[1], c, 1)
        self.latlayer2 = nn.Conv2d(in_channels[0], c, 1)
        self.semantic_branch = nn.Conv2d(c, c // 2, 3, 1, 1)
        self.semantic_branch2 = nn.Conv2d(c, c // 2, 3, 1, 1)
        self.conv2 = nn.Conv2d(c, c, 3, 1, 1)
        self.conv3 = nn.Conv2d(c, c, 3, 1, 1)
        self.feature_layer = nn.Conv2d(c // 2, 8, 1)
        self.output_layer_binary_seg = nn.Conv2d(8, 3, 1)
        self.output_layer_endp = nn.Conv2d(c // 2, 1, 1)
        self.gn11 = nn.GroupNorm(c // 2, c // 2)
        self.gn12 = nn.GroupNorm(c, c)
        self.gn21 = nn.GroupNorm(c // 2, c // 2)
        self.gn22 = nn.GroupNorm(c, c)

    # -------------------------------------------------------------------------------- packing
    def _pack(self):
        P = {}
        P['stem_w'] = self.conv1.weight.permute(2, 3, 1, 0).contiguous().float()        # [7,7,3,64]
        P['stem_s'], P['stem_b'] = ops.fold_bn(self.bn1)
        for lname in ('layer1', 'layer2', 'layer3'):
            for i, blk in enumerate(getattr(self, lname)):
                k = f'{lname}.{i}'
                P[k + '.w1'] = ops.pack_mfma(blk.conv1.weight)
                P[k + '.s1'], P[k + '.b1'] = ops.fold_bn(blk.bn1)
                P[k + '.w2'] = ops.pack_mfma(blk.conv2.weight)
                P[k + '.s2'], P[k + '.b2'] = ops.fold_bn(blk.bn2)
                for q, conv in (('.w1q', blk.conv1), ('.w2q', blk.conv2)):
                    if _f44_ok(conv):
                        P[k + q] = _frag44(conv.weight)
                if blk.downsample is not None:
                    P[k + '.wd'] = ops.pack_mfma(blk.downsample[0].weight)
                    P[k + '.sd'], P[k + '.bd'] = ops.fold_bn(blk.downsample[1])
        for name in ('toplayer', 'smooth1', 'smooth2', 'smooth3', 'latlayer1', 'latlayer2', 'semantic_branch',
                     'semantic_branch2', 'conv2', 'conv3'):
            m = getattr(self, name)
            P[name + '.w'] = ops.pack_mfma(m.weight)
            P[name + '.b'] = m.bias.float().contiguous()
            if _f44_ok(m):
                P[name + '.wq'] = _frag44(m.weight)
        # the two branches convolve p2 and p3 with different weights: one launch with the output channels concatenated transforms each
        # input patch once per 4 N tiles instead of per 2 (same values per channel: an output column does not depend on its neighbours)
        a, b2 = self.semantic_branch, self.semantic_branch2
        if (MERGE_BRANCH_CONVS and 'semantic_branch.wq' in P and 'semantic_branch2.wq' in P and a.out_channels == b2.out_channels
                and self.gn11.eps == self.gn21.eps):
            P['semantic_branch_ab.wq'] = _frag44(torch.cat([a.weight, b2.weight], dim=0))
            P['semantic_branch_ab.b'] = torch.cat([a.bias, b2.bias]).float().contiguous()
            if ('conv2.wq' in P and 'conv3.wq' in P and self.conv2.out_channels == self.conv3.out_channels
                    and self.gn12.eps == self.gn22.eps):           # likewise conv2 / conv3 on p4
                P['conv23.wq'] = _frag44(torch.cat([self.conv2.weight, self.conv3.weight], dim=0))
                P['conv23.b'] = torch.cat([self.conv2.bias, self.conv3.bias]).float().contiguous()
        if self.out is not None:
            P['out.w'] = ops.pack_mfma(self.out.weight)
        for name in ('feature_layer', 'output_layer_binary_seg', 'output_layer_endp'):
            m = getattr(self, name)
            P[name + '.w'] = ops.pack_small(m.weight)
            P[name + '.b'] = m.bias.float().contiguous()
        for name in ('gn11', 'gn12', 'gn21', 'gn22'):
            m = getattr(self, name)
            P[name + '.g'], P[name + '.b'] = m.weight.float().contiguous(), m.bias.float().contiguous()
        return P

    # -------------------------------------------------------------------------------- forward
    @staticmethod
    def _c3(x, P, wkey, cout, stride, dil, **epi):
        """3x3 convolution, pad = dilation: Winograd F(4x4,3x3) when its fragments were packed and the shape is covered, else the
        direct kernel."""
        if (wkey + 'q') in P and stride == 1 and ops.wino44_supported(x.shape[2], x.shape[3], x.shape[1], dil):
            return ops.conv_wino44(x, P[wkey + 'q'], cout, dil, **epi)
        return ops.conv_mfma(x, P[wkey], cout, 3, 3, stride, dil, dil, **epi)

    def _block(self, x, P, key, blk):
        cout = blk.conv1.out_channels
        y = self._c3(x, P, key + '.w1', cout, blk.stride, blk.dilation, scale=P[key + '.s1'], shift=P[key + '.b1'], act=ops.ACT_RELU)
        if blk.downsample is not None:
            x = ops.conv_mfma(x, P[key + '.wd'], cout, 1, 1, blk.stride, 0, 1, scale=P[key + '.sd'], shift=P[key + '.bd'])
        return self._c3(y, P, key + '.w2', cout, 1, blk.dilation, scale=P[key + '.s2'], shift=P[key + '.b2'], res=x, act=ops.ACT_RELU)

    def _conv3(self, x, P, name, cout):
        return self._c3(x, P, name + '.w', cout, 1, 1, shift=P[name + '.b'])

    def _conv_stats(self, P, src, conv, cout, gn):
        """conv3x3 + bias with the GroupNorm statistics coming out of the conv epilogue."""
        eps = getattr(self, gn).eps
        if (conv + '.wq') in P and ops.wino44_supported(src.shape[2], src.shape[3], src.shape[1], 1):
            return ops.conv_wino44(src, P[conv + '.wq'], cout, 1, shift=P[conv + '.b'], gn_eps=eps)
        if (src.shape[2] * src.shape[3]) % 128 == 0:      # whole 128-row tiles per image: statistics from the direct kernel's epilogue
            return ops.conv_mfma_gnstats(src, P[conv + '.w'], cout, 3, 3, 1, 1, 1, P[conv + '.b'], eps)
        t = self._conv3(src, P, conv, cout)               # ragged image size: separate statistics kernel
        return t, ops.gn_stats(t, eps)

    def _semantic(self, P, p2, p3, p4, conv_a, gn_a, conv_b, gn_b, proj, pre=None, pre_a4=None):
        """One of the two branches (reference :615-621 / :641-647): s2 + s3 + s4 at p2's size, followed by the branch's 1x1 output
        layer `proj` = (packed weight, bias, cout, out) - the only consumer of the sum, so the 128-channel sum is never written
        (lm_gn_relu_upsample_sum_conv1x1).  pre / pre_a4: (tensor, statistics) pairs of the merged launches (channel slices)."""
        h, w = p2.shape[2:]
        c_half = self.semantic_branch.out_channels
        t, st = pre_a4 if pre_a4 is not None else self._conv_stats(P, p4, conv_a, p4.shape[1], gn_a)
        s4 = ops.gn_relu_upsample_sum([(t, st)], P[gn_a + '.g'], P[gn_a + '.b'], (h, w))       # 256 ch at 288^2 (t may be a channel slice)
        terms = [pre[0] if pre is not None and pre[0] is not None else self._conv_stats(P, p2, conv_b, c_half, gn_b),       # s2
                 pre[1] if pre is not None and pre[1] is not None else self._conv_stats(P, p3, conv_b, c_half, gn_b)]       # s3
        terms.append(self._conv_stats(P, s4, conv_b, c_half, gn_b))                             # s4
        # (s2 + s3) + s4, each term GN + ReLU + bilinear to p2's size, and the 1x1 output layer, in one pass
        return ops.gn_relu_upsample_sum(terms, P[gn_b + '.g'], P[gn_b + '.b'], (h, w), proj=proj, keep_sum=False)

    def forward(self, x, fea_up_out=None):
        """Goes through the dispatcher: torch.ops.lanemap_hip.fpn_encoder (torch_ops.py; schema + fake kernel, cuda-only kernel)."""
        from . import torch_ops
        if self.out is None:
            return self._forward_impl(x, fea_up_out)
        if fea_up_out is None:
            B, H, W = (x.shape[0], x.shape[1], x.shape[2]) if x.dtype == torch.uint8 else (x.shape[0], x.shape[2], x.shape[3])
            fea_up_out = ops.new_act(B, 8, (H + 3) // 4, (W + 3) // 4, x.device)
        fea, bi_seg, endp = torch_ops.fpn_encoder(x, fea_up_out, torch_ops.stage_weights(self), torch_ops.stage_name(self))
        return fea, fea_up_out, bi_seg, endp

    def _forward_impl(self, x, fea_up_out=None):
        P = self.packed()
        if x.dtype == torch.uint8:          # u8 HWC tile straight from the rasteriser / PNG reader (ops.stem applies u8 / 255)
            B, H, W, _ = x.shape
        else:
            B, _, H, W = x.shape
        with trace.stage('fpn.stem'):
            c1 = ops.maxpool3x3s2(ops.stem(x, P['stem_w'], P['stem_s'], P['stem_b']))
        feats = []
        t = c1
        for lname in ('layer1', 'layer2', 'layer3'):
            with trace.stage('fpn.' + lname):
                for i, blk in enumerate(getattr(self, lname)):
                    t = self._block(t, P, f'{lname}.{i}', blk)
            feats.append(t)
        c2, c3, c4 = feats
        trace.push('fpn.topdown')
        fea = ops.conv_mfma(c4, P['out.w'], self.out.out_channels) if self.out is not None else None
        p4 = ops.conv_mfma(c4, P['toplayer.w'], 256, shift=P['toplayer.b'])
        # _upsample_add: the coarse map enters the lateral 1x1 convolution's epilogue through bilinear interpolation (a plain residual
        # when the sizes agree, as for p4 -> c3 under the dilated layer3)
        def lateral(c, name, coarse):
            if tuple(coarse.shape[2:]) == tuple(c.shape[2:]):
                return ops.conv_mfma(c, P[name + '.w'], 256, shift=P[name + '.b'], res=coarse)
            return ops.conv_mfma(c, P[name + '.w'], 256, shift=P[name + '.b'], res_up=coarse)

        p3 = lateral(c3, 'latlayer1', p4)
        p2 = lateral(c2, 'latlayer2', p3)
        del c1, c2, c3, c4, feats, t
        p4 = self._conv3(p4, P, 'smooth1', 256)
        p3 = self._conv3(p3, P, 'smooth2', 256)
        p2 = self._conv3(p2, P, 'smooth3', 256)
        trace.pop()
        trace.push('fpn.semantic')
        pre_a, pre_b = [None, None], [None, None]
        if 'semantic_branch_ab.wq' in P:
            ch = self.semantic_branch.out_channels
            for i, src in enumerate((p2, p3)):
                if ops.wino44_supported(src.shape[2], src.shape[3], src.shape[1], 1):
                    t, st = ops.conv_wino44(src, P['semantic_branch_ab.wq'], 2 * ch, 1, shift=P['semantic_branch_ab.b'],
                                            gn_eps=self.gn11.eps, gn_split=2)              # statistics per branch half: [2, B, ch, 2]
                    pre_a[i], pre_b[i] = (t[:, :ch], st[0]), (t[:, ch:], st[1])
        a4 = b4 = None
        if 'conv23.wq' in P and ops.wino44_supported(p4.shape[2], p4.shape[3], p4.shape[1], 1):
            c4o = self.conv2.out_channels
            t, st = ops.conv_wino44(p4, P['conv23.wq'], 2 * c4o, 1, shift=P['conv23.b'], gn_eps=self.gn12.eps, gn_split=2)
            a4, b4 = (t[:, :c4o], st[0]), (t[:, c4o:], st[1])
        fea_up = self._semantic(P, p2, p3, p4, 'conv2', 'gn12', 'semantic_branch', 'gn11',
                                (P['feature_layer.w'], P['feature_layer.b'], 8, fea_up_out), pre_a, a4)
        seg288 = ops.conv_small(fea_up, P['output_layer_binary_seg.w'], 3, shift=P['output_layer_binary_seg.b'], pre_relu=True)
        bi_seg = ops.upsample_to_chw(seg288, (H, W))
        endp288 = self._semantic(P, p2, p3, p4, 'conv3', 'gn22', 'semantic_branch2', 'gn21',
                                 (P['output_layer_endp.w'], P['output_layer_endp.b'], 1, None), pre_b, b4)
        endp = ops.upsample_to_chw(endp288, (H, W))
        trace.pop()
        return fea, fea_up, bi_seg, endp


@PCENCODER.register_module
class PostProjector2(nn.Module):
    def __init__(self, resnet='resnet50', pretrained=False, replace_stride_with_dilation=[False, True, False],
                 out_conv=True, in_channels=[64, 128, 256, -1], cfg=None):
        super().__init__()
        self.cfg = cfg
        self.fpn = FPNEncoder(resnet=resnet, pretrained=pretrained, replace_stride_with_dilation=replace_stride_with_dilation,
                              out_conv=out_conv, in_channels=in_channels, cfg=cfg)

    def forward(self, sample):
        return self.fpn(sample['proj'])

    def infer_validate(self, preds, seg_thre=None, endp_thre=None, display=None):
        """Segmentor decode (reference :115-183): raw-logit thresholds + clustered endpoint peaks."""
        from .decode import segmentor_decode
        return segmentor_decode(preds['seg'], preds['endp'], seg_thre)
