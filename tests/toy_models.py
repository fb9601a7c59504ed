"""Tiny stand-ins with the attribute layout of the HF Qwen2-VL and InternVL2 modules that the
rotation passes walk (fake_quant/qwen2vl_rotation.py, fake_quant/internvl_rotation.py).  The
forwards are plain attention/MLP stacks (no rotary, no masks): enough to check that LayerNorm
fusion + rotation leave the network function unchanged.  ``online_visual`` / ``online_llm`` make
the forward apply the run-time Hadamard in front of fc2 / down_proj (w2), as the quantised model
does through ActQuantWrapper."""
import types

import torch
import torch.nn as nn
import torch.nn.functional as F


class RMSNorm(nn.Module):                      # like Qwen2RMSNorm / InternLM2RMSNorm: no `bias` attribute
    def __init__(self, dim, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.eps = eps

    def forward(self, x):
        return self.weight * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + self.eps))


def _had(x, n_to):
    from fake_quant.hadamard_utils import matmul_hadU
    if x.shape[-1] != n_to:
        x = F.pad(x, (0, n_to - x.shape[-1]))
    return matmul_hadU(x)


def _attend(q, k, v):
    """q,k,v: [T, heads, d]"""
    w = torch.softmax(torch.einsum("thd,shd->hts", q, k) / q.shape[-1] ** 0.5, dim=-1)
    return torch.einsum("hts,shd->thd", w, v)


class VisAttn(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.qkv = nn.Linear(dim, 3 * dim, bias=True)
        self.proj = nn.Linear(dim, dim, bias=True)

    def forward(self, x):
        T, D = x.shape
        q, k, v = self.qkv(x).view(T, 3, self.num_heads, -1).unbind(1)
        return self.proj(_attend(q, k, v).reshape(T, D))


class VisMlp(nn.Module):
    def __init__(self, dim, hidden, owner):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden, bias=True)
        self.fc2 = nn.Linear(hidden, dim, bias=True)
        self.owner = owner

    def forward(self, x):
        h = F.gelu(self.fc1(x))
        if self.owner[0].online_visual:
            h = _had(h, self.fc2.in_features)
        return self.fc2(h)


class VisBlock(nn.Module):
    def __init__(self, dim, heads, hidden, owner):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = VisAttn(dim, heads)
        self.mlp = VisMlp(dim, hidden, owner)

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


# ------------------------------------------------------------------------------------ Qwen2-VL
class QwenLlmAttn(nn.Module):
    def __init__(self, dim, heads, kv_heads):
        super().__init__()
        self.heads, self.kv_heads, self.hd = heads, kv_heads, dim // heads
        self.q_proj = nn.Linear(dim, dim, bias=True)
        self.k_proj = nn.Linear(dim, kv_heads * self.hd, bias=True)
        self.v_proj = nn.Linear(dim, kv_heads * self.hd, bias=True)
        self.o_proj = nn.Linear(dim, dim, bias=False)

    def forward(self, x):
        T = x.shape[0]
        rep = self.heads // self.kv_heads
        q = self.q_proj(x).view(T, self.heads, self.hd)
        k = self.k_proj(x).view(T, self.kv_heads, self.hd).repeat_interleave(rep, dim=1)
        v = self.v_proj(x).view(T, self.kv_heads, self.hd).repeat_interleave(rep, dim=1)
        return self.o_proj(_attend(q, k, v).reshape(T, -1))


class QwenLlmMlp(nn.Module):
    def __init__(self, dim, inter, owner):
        super().__init__()
        self.gate_proj = nn.Linear(dim, inter, bias=False)
        self.up_proj = nn.Linear(dim, inter, bias=False)
        self.down_proj = nn.Linear(inter, dim, bias=False)
        self.owner = owner

    def forward(self, x):
        h = F.silu(self.gate_proj(x)) * self.up_proj(x)
        if self.owner[0].online_llm:
            h = _had(h, self.down_proj.in_features)
        return self.down_proj(h)


class QwenLlmLayer(nn.Module):
    def __init__(self, dim, heads, kv_heads, inter, owner):
        super().__init__()
        self.input_layernorm = RMSNorm(dim)
        self.post_attention_layernorm = RMSNorm(dim)
        self.self_attn = QwenLlmAttn(dim, heads, kv_heads)
        self.mlp = QwenLlmMlp(dim, inter, owner)

    def forward(self, x):
        x = x + self.self_attn(self.input_layernorm(x))
        return x + self.mlp(self.post_attention_layernorm(x))


class QwenPatchEmbed(nn.Module):
    def __init__(self, patch, vdim):
        super().__init__()
        self.proj = nn.Conv3d(patch[0], vdim, kernel_size=patch[1:], stride=patch[1:], bias=False)
        self.embed_dim = vdim

    def forward(self, pixels):
        return self.proj(pixels).view(pixels.shape[0], -1)


class QwenMerger(nn.Module):
    def __init__(self, vdim, dim):
        super().__init__()
        self.ln_q = nn.LayerNorm(vdim, eps=1e-6)
        self.mlp = nn.Sequential(nn.Linear(4 * vdim, 4 * vdim), nn.GELU(), nn.Linear(4 * vdim, dim))

    def forward(self, x):
        return self.mlp(self.ln_q(x).view(-1, 4 * x.shape[-1]))


class ToyQwen2VL(nn.Module):
    def __init__(self, vdim=32, vheads=2, vhidden=48, vdepth=2, dim=64, heads=4, kv_heads=2, inter=96,
                 depth=2, vocab=50, patch=(3, 2, 4, 4)):
        super().__init__()
        self.online_visual = self.online_llm = False
        owner = [self]
        vis = nn.Module()
        vis.patch_embed = QwenPatchEmbed(patch, vdim)
        vis.blocks = nn.ModuleList(VisBlock(vdim, vheads, vhidden, owner) for _ in range(vdepth))
        vis.merger = QwenMerger(vdim, dim)
        self.visual = vis
        self.model = nn.Module()
        self.model.embed_tokens = nn.Embedding(vocab, dim)
        self.model.layers = nn.ModuleList(QwenLlmLayer(dim, heads, kv_heads, inter, owner) for _ in range(depth))
        self.model.norm = RMSNorm(dim)
        self.lm_head = nn.Linear(dim, vocab, bias=False)
        self.config = types.SimpleNamespace(hidden_size=dim, num_attention_heads=heads,
                                            num_key_value_heads=kv_heads, intermediate_size=inter)
        self.patch = patch
        for p in self.parameters():             # non-trivial norm affine parts, biases and embeddings
            if p.dim() == 1:
                p.data = torch.randn_like(p) * 0.3 + (1.0 if p.numel() in (vdim, dim) else 0.0)

    def forward(self, pixels, ids):
        """pixels [P, C, T, H, W] (P % 4 == 0), ids [S] -> logits [P/4 + S, vocab]"""
        v = self.visual
        x = v.patch_embed(pixels)
        for blk in v.blocks:
            x = blk(x)
        x = v.merger(x)
        h = torch.cat([x, self.model.embed_tokens(ids)], 0)
        for layer in self.model.layers:
            h = layer(h)
        return self.lm_head(self.model.norm(h))


# ------------------------------------------------------------------------------------ InternVL2
class InternLlmAttn(nn.Module):
    def __init__(self, dim, heads, kv_heads):
        super().__init__()
        self.heads, self.kv_heads, self.hd = heads, kv_heads, dim // heads
        self.wqkv = nn.Linear(dim, (heads + 2 * kv_heads) * self.hd, bias=False)
        self.wo = nn.Linear(dim, dim, bias=False)

    def forward(self, x):
        T = x.shape[0]
        g = self.heads // self.kv_heads
        qkv = self.wqkv(x).view(T, self.kv_heads, 2 + g, self.hd)
        q = qkv[:, :, :g].reshape(T, self.heads, self.hd)
        k = qkv[:, :, -2].repeat_interleave(g, dim=1)
        v = qkv[:, :, -1].repeat_interleave(g, dim=1)
        return self.wo(_attend(q, k, v).reshape(T, -1))


class InternLlmMlp(nn.Module):
    def __init__(self, dim, inter, owner):
        super().__init__()
        self.w1 = nn.Linear(dim, inter, bias=False)
        self.w3 = nn.Linear(dim, inter, bias=False)
        self.w2 = nn.Linear(inter, dim, bias=False)
        self.owner = owner

    def forward(self, x):
        h = F.silu(self.w1(x)) * self.w3(x)
        if self.owner[0].online_llm:
            h = _had(h, self.w2.in_features)
        return self.w2(h)


class InternLlmLayer(nn.Module):
    def __init__(self, dim, heads, kv_heads, inter, owner):
        super().__init__()
        self.attention_norm = RMSNorm(dim)
        self.ffn_norm = RMSNorm(dim)
        self.attention = InternLlmAttn(dim, heads, kv_heads)
        self.feed_forward = InternLlmMlp(dim, inter, owner)

    def forward(self, x):
        x = x + self.attention(self.attention_norm(x))
        return x + self.feed_forward(self.ffn_norm(x))


class ToyInternVL(nn.Module):
    def __init__(self, vdim=32, vheads=2, vhidden=48, vdepth=2, dim=64, heads=4, kv_heads=2, inter=96,
                 depth=2, vocab=50, patches=8, patch=(3, 4, 4)):
        super().__init__()
        self.online_visual = self.online_llm = False
        owner = [self]
        vm = nn.Module()
        vm.embeddings = nn.Module()
        vm.embeddings.patch_embedding = nn.Conv2d(patch[0], vdim, kernel_size=patch[1:], stride=patch[1:], bias=True)
        vm.embeddings.class_embedding = nn.Parameter(torch.randn(1, 1, vdim))
        vm.embeddings.position_embedding = nn.Parameter(torch.randn(1, patches + 1, vdim))
        vm.encoder = nn.Module()
        vm.encoder.layers = nn.ModuleList(VisBlock(vdim, vheads, vhidden, owner) for _ in range(vdepth))
        vm.encoder.config = types.SimpleNamespace(hidden_size=vdim)
        self.vision_model = vm
        self.mlp1 = nn.Sequential(nn.LayerNorm(4 * vdim), nn.Linear(4 * vdim, dim), nn.GELU(), nn.Linear(dim, dim))
        lm = nn.Module()
        lm.model = nn.Module()
        lm.model.tok_embeddings = nn.Embedding(vocab, dim)
        lm.model.layers = nn.ModuleList(InternLlmLayer(dim, heads, kv_heads, inter, owner) for _ in range(depth))
        lm.model.norm = RMSNorm(dim)
        lm.output = nn.Linear(dim, vocab, bias=False)
        self.language_model = lm
        self.config = types.SimpleNamespace(
            downsample_ratio=0.5,
            vision_config=types.SimpleNamespace(num_attention_heads=vheads, hidden_size=vdim),
            llm_config=types.SimpleNamespace(hidden_size=dim, num_attention_heads=heads, num_key_value_heads=kv_heads))
        for p in self.parameters():
            if p.dim() == 1:
                p.data = torch.randn_like(p) * 0.3 + (1.0 if p.numel() in (vdim, dim, 4 * vdim) else 0.0)

    def forward(self, pixels, ids):
        """pixels [P, C, H, W] (one image of P patches), ids [S] -> logits [P/4 + S, vocab]"""
        e = self.vision_model.embeddings
        x = e.patch_embedding(pixels).view(1, pixels.shape[0], -1)
        x = (torch.cat([e.class_embedding, x], 1) + e.position_embedding)[0]
        for layer in self.vision_model.encoder.layers:
            x = layer(x)
        x = self.mlp1(x[1:].reshape(-1, 4 * x.shape[-1]))      # drop cls, 2x2 "pixel shuffle"
        h = torch.cat([x, self.language_model.model.tok_embeddings(ids)], 0)
        for layer in self.language_model.model.layers:
            h = layer(h)
        return self.language_model.output(self.language_model.model.norm(h))


# ------------------------------------------------------------------------------------ Qwen-VL (v1, "opt" layout)
class MHA(nn.Module):
    """q/k/v/out Linears as in visual_opt.VisualAttention after the q/k/v split."""
    def __init__(self, dim, heads, out_name="out_proj", bias=True, out_bias=True):
        super().__init__()
        self.heads = heads
        self.q_proj = nn.Linear(dim, dim, bias=bias)
        self.k_proj = nn.Linear(dim, dim, bias=bias)
        self.v_proj = nn.Linear(dim, dim, bias=bias)
        setattr(self, out_name, nn.Linear(dim, dim, bias=out_bias))
        self.out_name = out_name

    def forward(self, q_in, k_in, v_in):
        h = self.heads
        q = self.q_proj(q_in).view(q_in.shape[0], h, -1)
        k = self.k_proj(k_in).view(k_in.shape[0], h, -1)
        v = self.v_proj(v_in).view(v_in.shape[0], h, -1)
        return getattr(self, self.out_name)(_attend(q, k, v).reshape(q_in.shape[0], -1))


class QwenVisMlp(nn.Module):
    def __init__(self, dim, hidden, owner):
        super().__init__()
        self.c_fc = nn.Linear(dim, hidden)
        self.c_proj = nn.Linear(hidden, dim)
        self.owner = owner

    def forward(self, x):
        h = F.gelu(self.c_fc(x))
        if self.owner[0].online_visual:
            h = _had(h, self.c_proj.in_features)
        return self.c_proj(h)


class QwenVisBlock(nn.Module):
    def __init__(self, dim, heads, hidden, owner):
        super().__init__()
        self.ln_1 = nn.LayerNorm(dim, eps=1e-6)
        self.ln_2 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = MHA(dim, heads)
        self.mlp = QwenVisMlp(dim, hidden, owner)

    def forward(self, x):
        y = self.ln_1(x)
        x = x + self.attn(y, y, y)
        return x + self.mlp(self.ln_2(x))


class QwenResampler(nn.Module):
    def __init__(self, width, dim, heads, queries, patches):
        super().__init__()
        self.embed_dim, self.num_heads = dim, heads
        self.kv_proj = nn.Linear(width, dim, bias=False)
        self.ln_kv = nn.LayerNorm(dim, eps=1e-6)
        self.ln_q = nn.LayerNorm(dim, eps=1e-6)
        self.query = nn.Parameter(torch.randn(queries, dim))
        self.pos_embed = nn.Parameter(torch.randn(queries, dim))
        self.pos_embed_kv = nn.Parameter(torch.randn(patches, dim))
        self.attn = MHA(dim, heads)

    def forward(self, x):
        kv = self.ln_kv(self.kv_proj(x))
        return self.attn(self.ln_q(self.query) + self.pos_embed, kv + self.pos_embed_kv, kv)


class QwenV1Mlp(nn.Module):
    def __init__(self, dim, inter, owner):
        super().__init__()
        self.w1 = nn.Linear(dim, inter, bias=False)
        self.w2 = nn.Linear(dim, inter, bias=False)
        self.c_proj = nn.Linear(inter, dim, bias=False)
        self.owner = owner

    def forward(self, x):
        h = self.w1(x) * F.silu(self.w2(x))
        if self.owner[0].online_llm:
            h = _had(h, self.c_proj.in_features)
        return self.c_proj(h)


class QwenV1Block(nn.Module):
    def __init__(self, dim, heads, inter, owner):
        super().__init__()
        self.ln_1 = RMSNorm(dim)
        self.ln_2 = RMSNorm(dim)
        self.attn = MHA(dim, heads, out_name="c_proj", bias=True, out_bias=False)
        self.mlp = QwenV1Mlp(dim, inter, owner)

    def forward(self, x):
        y = self.ln_1(x)
        x = x + self.attn(y, y, y)
        return x + self.mlp(self.ln_2(x))


class ToyQwenVL(nn.Module):
    def __init__(self, width=32, vheads=2, vhidden=48, vdepth=2, dim=64, heads=4, inter=96, depth=2,
                 vocab=50, patches=8, queries=4, patch_dim=24):
        super().__init__()
        self.online_visual = self.online_llm = False
        owner = [self]
        vis = nn.Module()
        vis.conv1 = nn.Linear(patch_dim, width, bias=False)          # stands in for the patch conv
        vis.positional_embedding = nn.Parameter(torch.randn(patches, width))
        vis.ln_pre = nn.LayerNorm(width, eps=1e-6)
        vis.fc_sub_mean = nn.Linear(width, width, bias=False)
        vis.fc_sub_mean.weight.data = torch.eye(width) - 1.0 / width
        vis.transformer = nn.Module()
        vis.transformer.resblocks = nn.ModuleList(QwenVisBlock(width, vheads, vhidden, owner) for _ in range(vdepth))
        vis.attn_pool = QwenResampler(width, dim, heads, queries, patches)
        vis.ln_post = nn.LayerNorm(dim, eps=1e-6)
        vis.proj_fc = nn.Linear(dim, dim, bias=True)
        self.transformer = nn.Module()
        self.transformer.visual = vis
        self.transformer.wte = nn.Embedding(vocab, dim)
        self.transformer.h = nn.ModuleList(QwenV1Block(dim, heads, inter, owner) for _ in range(depth))
        self.transformer.ln_f = RMSNorm(dim)
        self.lm_head = nn.Linear(dim, vocab, bias=False)
        self.config = types.SimpleNamespace(visual=dict(heads=vheads, width=width, output_dim=dim), hidden_size=dim,
                                            num_attention_heads=heads, intermediate_size=inter)
        for p in self.parameters():
            if p.dim() == 1:
                p.data = torch.randn_like(p) * 0.3 + (1.0 if p.numel() in (width, dim) else 0.0)

    def forward(self, pixels, ids):
        """pixels [patches, patch_dim], ids [S] -> logits [queries + S, vocab]"""
        v = self.transformer.visual
        x = v.fc_sub_mean(v.ln_pre(v.conv1(pixels) + v.positional_embedding))
        for blk in v.transformer.resblocks:
            x = blk(x)
        x = v.proj_fc(v.ln_post(v.attn_pool(x)))
        h = torch.cat([x, self.transformer.wte(ids)], 0)
        for layer in self.transformer.h:
            h = layer(h)
        return self.lm_head(self.transformer.ln_f(h))


# ------------------------------------------------------------------------------------ MiniCPM-V
class SiglipLayer(nn.Module):
    def __init__(self, dim, heads, hidden, owner):
        super().__init__()
        self.layer_norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.layer_norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.self_attn = MHA(dim, heads)
        self.self_attn.num_heads, self.self_attn.head_dim = heads, dim // heads
        self.mlp = nn.Module()
        self.mlp.fc1 = nn.Linear(dim, hidden)
        self.mlp.fc2 = nn.Linear(hidden, dim)
        self.owner = owner

    def forward(self, x):
        y = self.layer_norm1(x)
        x = x + self.self_attn(y, y, y)
        h = F.gelu(self.mlp.fc1(self.layer_norm2(x)))
        if self.owner[0].online_visual:
            h = _had(h, self.mlp.fc2.in_features)
        return x + self.mlp.fc2(h)


class MiniCpmResampler(nn.Module):
    def __init__(self, width, dim, heads, queries, patches):
        super().__init__()
        self.embed_dim, self.num_heads = dim, heads
        self.kv_proj = nn.Linear(width, dim, bias=False)
        self.ln_kv, self.ln_q, self.ln_post = nn.LayerNorm(dim, eps=1e-6), nn.LayerNorm(dim, eps=1e-6), nn.LayerNorm(dim, eps=1e-6)
        self.query = nn.Parameter(torch.randn(queries, dim))
        self.pos_embed = nn.Parameter(torch.randn(patches, dim))
        self.attn = MHA(dim, heads)
        self.proj_fc = nn.Linear(dim, dim, bias=True)

    def forward(self, x):
        kv = self.ln_kv(self.kv_proj(x))
        out = self.attn(self.ln_q(self.query), kv + self.pos_embed, kv)
        return self.proj_fc(self.ln_post(out))


class MiniCpmLlmLayer(nn.Module):
    def __init__(self, dim, heads, inter, owner):
        super().__init__()
        self.input_layernorm, self.post_attention_layernorm = RMSNorm(dim), RMSNorm(dim)
        self.self_attn = MHA(dim, heads, out_name="o_proj", bias=False, out_bias=False)
        self.mlp = QwenLlmMlp(dim, inter, owner)

    def forward(self, x):
        y = self.input_layernorm(x)
        x = x + self.self_attn(y, y, y)
        return x + self.mlp(self.post_attention_layernorm(x))


class ToyMiniCPMV(nn.Module):
    def __init__(self, width=32, vheads=2, vhidden=48, vdepth=2, dim=64, heads=4, inter=96, depth=2, vocab=50,
                 patches=8, queries=4):
        super().__init__()
        self.online_visual = self.online_llm = False
        owner = [self]
        self.vpm = nn.Module()
        self.vpm.embed_dim = width
        self.vpm.embeddings = nn.Module()
        self.vpm.embeddings.embed_dim = width
        self.vpm.embeddings.patch_embedding = nn.Conv2d(3, width, kernel_size=4, stride=4, bias=True)
        self.vpm.embeddings.position_embedding = nn.Embedding(patches, width)
        self.vpm.encoder = nn.Module()
        self.vpm.encoder.layers = nn.ModuleList(SiglipLayer(width, vheads, vhidden, owner) for _ in range(vdepth))
        self.vpm.post_layernorm = nn.LayerNorm(width, eps=1e-6)
        self.resampler = MiniCpmResampler(width, dim, heads, queries, patches)
        self.llm = nn.Module()
        self.llm.model = nn.Module()
        self.llm.model.embed_tokens = nn.Embedding(vocab, dim)
        self.llm.model.layers = nn.ModuleList(MiniCpmLlmLayer(dim, heads, inter, owner) for _ in range(depth))
        self.llm.model.norm = RMSNorm(dim)
        self.llm.lm_head = nn.Linear(dim, vocab, bias=False)
        self.config = types.SimpleNamespace(hidden_size=dim, num_attention_heads=heads, intermediate_size=inter,
                                            vision_config=types.SimpleNamespace(intermediate_size=vhidden))
        for p in self.parameters():
            if p.dim() == 1:
                p.data = torch.randn_like(p) * 0.3 + (1.0 if p.numel() in (width, dim) else 0.0)

    def forward(self, pixels, ids):
        """pixels [patches, 3, 4, 4], ids [S] -> logits [queries + S, vocab]"""
        e = self.vpm.embeddings
        x = e.patch_embedding(pixels).flatten(1) + e.position_embedding.weight
        for layer in self.vpm.encoder.layers:
            x = layer(x)
        x = self.resampler(self.vpm.post_layernorm(x))
        h = torch.cat([x, self.llm.model.embed_tokens(ids)], 0)
        for layer in self.llm.model.layers:
            h = layer(h)
        return self.llm.lm_head(self.llm.model.norm(h))


def rotation_args(**over):
    base = dict(no_fuse_visual_clip=False, no_fuse_visual_cross_attn=False, no_fuse_llm=False,
                rotate_visual_clip=True, rotate_visual_cross_attn=True, rotate_llm=True,
                rotate_mode="hadamard", online_visual_hadamard=True, online_llm_hadamard=True)
    base.update(over)
    return types.SimpleNamespace(**base)


def build(kind, seed, **kw):
    torch.manual_seed(seed)
    model = {"qwen2vl": ToyQwen2VL, "internvl": ToyInternVL, "qwenvl": ToyQwenVL,
             "minicpmv": ToyMiniCPMV}[kind](**kw).double().eval()
    g = torch.Generator().manual_seed(seed + 1)
    if kind == "qwen2vl":
        pixels = torch.randn(8, *model.patch, generator=g, dtype=torch.float64)
    elif kind == "qwenvl":
        pixels = torch.randn(8, 24, generator=g, dtype=torch.float64)
    elif kind == "minicpmv":
        pixels = torch.randn(8, 3, 4, 4, generator=g, dtype=torch.float64)
    else:
        pixels = torch.randn(8, 3, 4, 4, generator=g, dtype=torch.float64)
    ids = torch.randint(0, 50, (5,), generator=g)
    return model, pixels, ids
