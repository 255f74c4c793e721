"""GFC-T / ViT block behind the reference's BACKBONE registry name ``VitSegNet``.

Drop-in for baseline/models/backbone/vitsegnet.py:132-214 (same kwargs, same state-dict keys:
``to_patch_embedding.1``, ``pos_embedding``, ``transformer.layers.{l}.{0,1}.{norm,fn...}``).
Kernels: patch embedding = 8x8/stride-8 implicit GEMM straight from the NHWC feature map (no patchify
copy) with the positional embedding added in the epilogue; LayerNorm rows; QKV / out-proj / MLP GEMMs
on lm_conv2d_nhwc_mfma_f32 with bias + residual + erf-GELU epilogues; lm_attention_f32.
"""
import torch
import torch.nn as nn

from . import ops
from .registry import BACKBONE
from .packing import PackedModule


class _PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn


class _Attention(nn.Module):
    def __init__(self, dim, heads, dim_head):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.dim_head, self.scale = heads, dim_head, dim_head ** -0.5
        self.to_qkv = nn.Linear(dim, inner * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Dropout(0.)) if not (heads == 1 and dim_head == dim) else nn.Identity()


class _FeedForward(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, hidden), nn.GELU(), nn.Dropout(0.), nn.Linear(hidden, dim), nn.Dropout(0.))


class _Transformer(nn.Module):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim):
        super().__init__()
        self.layers = nn.ModuleList([nn.ModuleList([_PreNorm(dim, _Attention(dim, heads, dim_head)),
                                                    _PreNorm(dim, _FeedForward(dim, mlp_dim))]) for _ in range(depth)])


def transformer_forward(layers, P, prefix, t, B, N, valid=None):
    """t [B*N, dim] -> [B*N, dim]; pre-norm blocks `x = attn(x) + x; x = ff(x) + x` (vitsegnet.py:79-83).  valid: key mask of the
    attention core (ops.attention); everything else is row-wise, rows of unflagged tokens are computed and ignored by the caller."""
    for l, (attn, ff) in enumerate(layers):
        k = f'{prefix}{l}'
        y = ops.layernorm(t, P[k + '.ln1.g'], P[k + '.ln1.b'], attn.norm.eps)
        qkv = ops.linear_mfma(y, P[k + '.qkv'], attn.fn.to_qkv.out_features)
        o = ops.attention(qkv, B, N, attn.fn.heads, attn.fn.dim_head, attn.fn.scale, valid=valid)
        t = ops.linear_mfma(o, P[k + '.proj'], t.shape[1], shift=P[k + '.proj.b'], res=t)
        y = ops.layernorm(t, P[k + '.ln2.g'], P[k + '.ln2.b'], ff.norm.eps)
        y = ops.linear_mfma(y, P[k + '.fc1'], ff.fn.net[0].out_features, shift=P[k + '.fc1.b'], act=ops.ACT_GELU)
        t = ops.linear_mfma(y, P[k + '.fc2'], t.shape[1], shift=P[k + '.fc2.b'], res=t)
    return t


def pack_transformer(layers, P, prefix):
    for l, (attn, ff) in enumerate(layers):
        k = f'{prefix}{l}'
        P[k + '.ln1.g'], P[k + '.ln1.b'] = attn.norm.weight.float().contiguous(), attn.norm.bias.float().contiguous()
        P[k + '.qkv'] = ops.pack_mfma(attn.fn.to_qkv.weight)
        P[k + '.proj'] = ops.pack_mfma(attn.fn.to_out[0].weight)
        P[k + '.proj.b'] = attn.fn.to_out[0].bias.float().contiguous()
        P[k + '.ln2.g'], P[k + '.ln2.b'] = ff.norm.weight.float().contiguous(), ff.norm.bias.float().contiguous()
        P[k + '.fc1'] = ops.pack_mfma(ff.fn.net[0].weight)
        P[k + '.fc1.b'] = ff.fn.net[0].bias.float().contiguous()
        P[k + '.fc2'] = ops.pack_mfma(ff.fn.net[3].weight)
        P[k + '.fc2.b'] = ff.fn.net[3].bias.float().contiguous()


@BACKBONE.register_module
class VitSegNet(PackedModule):
    def __init__(self, image_size=144, patch_h_size=8, patch_w_size=8, channels=64, dim=512, depth=5, heads=16,
                 output_channels=1024, expansion_factor=4, dim_head=64, dropout=0., emb_dropout=0.,
                 is_with_shared_mlp=True, is_with_llm=False, cfg=None):
        super().__init__()
        assert image_size % patch_h_size == 0 and image_size % patch_w_size == 0, \
            'Image dimensions must be divisible by the patch size.'
        if patch_h_size != patch_w_size:
            raise NotImplementedError('square patches only')
        self.patch, self.grid, self.channels, self.dim = patch_h_size, image_size // patch_h_size, channels, dim
        self.to_patch_embedding = nn.Sequential(nn.Identity(), nn.Linear(channels * patch_h_size * patch_w_size, dim))
        self.pos_embedding = nn.Parameter(torch.randn(1, self.grid * self.grid, dim))
        self.dropout = nn.Dropout(emb_dropout)
        self.transformer = _Transformer(dim, depth, heads, dim_head, int(dim * expansion_factor))
        self.out_c = dim // (patch_h_size * patch_w_size)
        self.is_with_shared_mlp = bool(is_with_shared_mlp)
        if self.is_with_shared_mlp:
            self.shared_mlp = nn.Conv2d(self.out_c, output_channels, 1)

    def _pack(self):
        P = {}
        lin = self.to_patch_embedding[1]
        p = self.patch
        # Linear weight columns are ordered (p1 p2 c) == (kh kw cin) of an 8x8/stride-8 conv over NHWC
        w = lin.weight.reshape(self.dim, p, p, self.channels).permute(0, 3, 1, 2)
        P['embed.w'] = ops.pack_mfma(w)
        P['embed.b'] = lin.bias.float().contiguous()
        P['pos'] = self.pos_embedding[0].float().contiguous()
        pack_transformer(self.transformer.layers, P, 'L')
        if self.is_with_shared_mlp:
            P['mlp.w'] = ops.pack_mfma(self.shared_mlp.weight)
            P['mlp.b'] = self.shared_mlp.bias.float().contiguous()
        return P

    def forward(self, img):
        """Goes through the dispatcher: torch.ops.lanemap_hip.vit_backbone (torch_ops.py)."""
        from . import torch_ops
        return torch_ops.vit_backbone(img, torch_ops.stage_weights(self), torch_ops.stage_name(self))

    def _forward_impl(self, img):
        P = self.packed()
        B = img.shape[0]
        N = self.grid * self.grid
        tok = ops.conv_mfma(img, P['embed.w'], self.dim, self.patch, self.patch, self.patch, 0, 1,
                            shift=P['embed.b'], res=P['pos'], res_rows=N)           # [B,dim,G,G] NHWC == [B*N, dim]
        t = tok.permute(0, 2, 3, 1).reshape(B * N, self.dim)
        t = transformer_forward(self.transformer.layers, P, 'L', t, B, N)
        x = ops.unpatchify(t, B, self.grid, self.patch, self.out_c)
        if self.is_with_shared_mlp:
            x = ops.conv_mfma(x, P['mlp.w'], self.shared_mlp.out_channels, shift=P['mlp.b'])
        return x
