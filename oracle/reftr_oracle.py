"""CPU oracle: a plain-torch fp32 restatement of the RefTR training hot path.

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
file; nothing under reftr_amd/ does, and the product path has no CPU fallback.

Every function restates one piece of the reference (ubc-vision/RefTR, /root/reference) and cites the
file:line it follows.  Parameters are a flat dict keyed by the REFERENCE's state_dict names, so the dict
of the imported reference model can be fed in directly: oracle/gen_golden.py does exactly that inside the
build container and pins this file against the reference's own outputs (tests/golden/*.npz, checked by
tests/test_oracle_golden.py).  Arithmetic that lives outside the reference tree (torchvision ResNet,
HF BertModel, nn.MultiheadAttention) is restated from its public definition — SURVEY.md Appendix A.

`q=True` turns on the bf16 rounding points of the HIP path (GEMM operands and stored backbone
activations rounded to bf16, FrozenBN scale folded into the rounded weight) so the GPU result can be
compared at the 1e-3 level the north star asks for; `q=False` is the reference's fp32 arithmetic.
"""
import math
from dataclasses import dataclass, field
from typing import Optional

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# configuration (defaults = main_vg.py:26-164 as used by configs/refcoco/RefTR_refcoco.sh)
# ----------------------------------------------------------------------------------------------
@dataclass
class BertCfg:
    vocab_size: int = 30522
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    inter: int = 3072
    max_pos: int = 512
    type_vocab: int = 2
    eps: float = 1e-12
    dropout: float = 0.1
    pad_idx: int = -1      # >= 0: RoBERTa position ids (HF create_position_ids_from_input_ids)


def roberta_cfg(layers=12):
    return BertCfg(vocab_size=50265, max_pos=514, type_vocab=1, eps=1e-5, pad_idx=1, layers=layers)


@dataclass
class Cfg:
    hidden: int = 256
    nheads: int = 8
    enc_layers: int = 6
    dec_layers: int = 6
    ffn: int = 2048
    dropout: float = 0.1
    max_lang_seq: int = 128
    n_q: int = 1
    aux_loss: bool = True
    resnet_layers: tuple = (3, 4, 6, 3)
    bert: BertCfg = field(default_factory=BertCfg)
    pos_learned: bool = False     # --position_embedding learned (position_encoding.py:59-84)
    dilation: bool = False        # --dilation: layer4's stride replaced by dilation (backbone.py:117-125, torchvision resnet._make_layer)
    masks: bool = False           # RefTRSeg (reftr_segmentation.py): RES head on top of the single-phrase REC model
    cem: bool = False             # --ablation cem_loss: the CEM block + loss_cem (reftr_segmentation.py:16-41, 62-64, 146-147)
    mask_loss_coef: float = 1.0   # main_vg.py (default 1)
    dice_loss_coef: float = 1.0
    bbox_loss_coef: float = 1.0   # main_vg.py:134 (default 1)
    giou_loss_coef: float = 1.0   # main_vg.py:135 (default 1)


# ----------------------------------------------------------------------------------------------
# Accumulation order of every contraction (conv / linear / bmm / matmul).  Default: torch's fp32 kernels (what the golden
# vectors were pinned with).  Inside `with accumulate_fp64():` the SAME operands are multiplied and summed in fp64 and the
# result is rounded to fp32 once -- another summation order of the same computation, with the same bf16 rounding points
# around it when q=True.  The distance between the two `q=True` forwards is the floor any bf16-operand implementation with its
# own accumulation order (the MFMA path) can be expected to sit at (tests/test_oracle_golden.py).
# ----------------------------------------------------------------------------------------------
_ACC = {"fp64": False, "fp32_trunk_from": 0, "perm": 0, "site": 0}      # fp32_trunk_from = L > 0: experiment, see `fp32_trunk`


class fp32_trunk:
    """Experiment knob (not a mode of the product): in q=True forwards, the residual trunk of ResNet stages >= `first_layer`
    (conv3 + identity outputs) is NOT rounded to bf16 when stored -- only re-rounded where it enters the next GEMM as an
    operand.  oracle/noise_floor.py uses it to show how much of the end-to-end distance the stored-trunk rounding carries."""

    def __init__(self, first_layer=3):
        self.first = first_layer

    def __enter__(self):
        self._old = _ACC["fp32_trunk_from"]
        _ACC["fp32_trunk_from"] = self.first

    def __exit__(self, *a):
        _ACC["fp32_trunk_from"] = self._old


class accumulate_fp64:
    def __enter__(self):
        self._old = _ACC["fp64"]
        _ACC["fp64"] = True

    def __exit__(self, *a):
        _ACC["fp64"] = self._old


class accumulate_permuted:
    """Yet another summation order of the same computation, in fp32: every contraction is cut into 2..5 chunks along its
    reduction axis and the chunk products are added in an order that depends on `seed` (and on the call site, so that no two
    contractions of a forward use the same cut).  Many seeds = many samples of the order floor (the
    gradient gates must come from a distribution of floor samples, not from one).  seed = 0: off."""

    def __init__(self, seed):
        self.seed = int(seed)

    def __enter__(self):
        self._old = (_ACC["perm"], _ACC["site"])
        _ACC["perm"], _ACC["site"] = self.seed, 0

    def __exit__(self, *a):
        _ACC["perm"], _ACC["site"] = self._old


def _perm_chunks(K):
    """[(begin, end), ...] of the reduction axis in the order the chunk products are added, or None (axis too short / off)."""
    seed = _ACC["perm"]
    if not seed or K < 32:
        return None
    _ACC["site"] += 1
    h = (seed * 2654435761 + _ACC["site"] * 40503) & 0xFFFFFFFF
    n = 2 + (h >> 3) % 4
    edges = [K * i // n for i in range(n + 1)]
    order = list(range(n))
    rot = (h >> 7) % n
    order = order[rot:] + order[:rot]
    if (h >> 11) & 1:
        order.reverse()
    return [(edges[i], edges[i + 1]) for i in order]


def _d(t):
    return None if t is None else t.double()


def conv2d_acc(x, w, b=None, *a, **k):
    if _ACC["fp64"]:
        return F.conv2d(_d(x), _d(w), _d(b), *a, **k).float()
    ch = _perm_chunks(x.shape[1]) if k.get("groups", 1) == 1 and (len(a) < 4 or a[3] == 1) else None
    if ch is not None:
        y = None
        for lo, hi in ch:
            part = F.conv2d(x[:, lo:hi], w[:, lo:hi], None, *a, **k)
            y = part if y is None else y + part
        return y if b is None else y + b.view(1, -1, 1, 1)
    return F.conv2d(x, w, b, *a, **k)


def linear_acc(x, w, b=None):
    if _ACC["fp64"]:
        return F.linear(_d(x), _d(w), _d(b)).float()
    ch = _perm_chunks(x.shape[-1])
    if ch is not None:
        y = None
        for lo, hi in ch:
            part = F.linear(x[..., lo:hi], w[:, lo:hi])
            y = part if y is None else y + part
        return y if b is None else y + b
    return F.linear(x, w, b)


def _mm_perm(fn, a, b):
    ch = _perm_chunks(a.shape[-1])
    if ch is None:
        return fn(a, b)
    y = None
    for lo, hi in ch:
        part = fn(a[..., lo:hi], b[..., lo:hi, :])
        y = part if y is None else y + part
    return y


def bmm_acc(a, b):
    return torch.bmm(a.double(), b.double()).float() if _ACC["fp64"] else _mm_perm(torch.bmm, a, b)


def matmul_acc(a, b):
    return torch.matmul(a.double(), b.double()).float() if _ACC["fp64"] else _mm_perm(torch.matmul, a, b)


class _Round(torch.autograd.Function):
    """bf16 rounding point, applied to the value going forward and to the gradient coming back."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


def rq(x, q):
    return _Round.apply(x) if q else x


def rq_fwd(x, q):
    """rounding of a weight / constant: forward only (the fp32 master weight receives the full gradient)."""
    return x + (x.to(torch.bfloat16).to(torch.float32) - x).detach() if q else x


# ----------------------------------------------------------------------------------------------
# backbone: torchvision ResNet v1.5 body + FrozenBatchNorm2d (models/modeling/backbone.py:43-125)
# ----------------------------------------------------------------------------------------------
def frozen_bn_affine(P, pfx, eps=1e-5):
    """backbone.py:70-80: scale = w * rsqrt(rv + eps); shift = b - rm * scale (all buffers, no grad)."""
    scale = P[pfx + "weight"] * (P[pfx + "running_var"] + eps).rsqrt()
    shift = P[pfx + "bias"] - P[pfx + "running_mean"] * scale
    return scale.detach(), shift.detach()


def conv_bn(x, P, conv, bn, stride=1, padding=0, q=False, relu=True, residual=None, round_out=True, dilation=1):
    w = P[conv + "weight"]
    scale, shift = frozen_bn_affine(P, bn)
    if q:   # HIP path: scale folded into the bf16 weight, output stored as bf16
        y = conv2d_acc(rq(x, q), rq_fwd(w * scale.view(-1, 1, 1, 1), q), None, stride, padding, dilation) + shift.view(1, -1, 1, 1)
    else:
        y = conv2d_acc(x, w, None, stride, padding, dilation) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    if residual is not None:
        y = y + residual
    if relu:
        y = F.relu(y)
    return rq(y, q and round_out)


def bottleneck(x, P, pfx, stride, q=False, trunk_round=True, dilation=1):
    """ResNet v1.5 bottleneck (stride on the 3x3; its padding = its dilation, torchvision conv3x3), SURVEY.md A1."""
    idt = x
    if (pfx + "downsample.0.weight") in P:
        idt = conv_bn(x, P, pfx + "downsample.0.", pfx + "downsample.1.", stride, 0, q, relu=False, round_out=trunk_round)
    y = conv_bn(x, P, pfx + "conv1.", pfx + "bn1.", 1, 0, q)
    y = conv_bn(y, P, pfx + "conv2.", pfx + "bn2.", stride, dilation, q, dilation=dilation)
    return conv_bn(y, P, pfx + "conv3.", pfx + "bn3.", 1, 0, q, relu=True, residual=idt, round_out=trunk_round)


def resnet_body(x, P, pfx="img_backbone.0.body.", layers=(3, 4, 6, 3), q=False, dilation=False):
    """conv1 7x7/2 + FrozenBN + ReLU + maxpool 3x3/2, then layer1..4 (backbone.py:99,119-121).
    Returns the four stage outputs (strides 4, 8, 16, 32).  dilation (backbone.py:117-125 -> torchvision's
    replace_stride_with_dilation=[False, False, True]): layer4's first block runs at stride 1 with dilation 1, its other blocks
    dilate their 3x3 by 2 -- the last output keeps stride 16."""
    scale, shift = frozen_bn_affine(P, pfx + "bn1.")
    w = P[pfx + "conv1.weight"]
    if q:
        y = conv2d_acc(rq(x, q), rq_fwd(w * scale.view(-1, 1, 1, 1), q), None, 2, 3) + shift.view(1, -1, 1, 1)
    else:
        y = conv2d_acc(x, w, None, 2, 3) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    y = rq(F.relu(y), q)
    y = F.max_pool2d(y, 3, 2, 1)
    outs = []
    for li, n in enumerate(layers):
        for bi in range(n):
            stride = 2 if (bi == 0 and li > 0) else 1
            dl = 1
            if dilation and li == 3:
                stride, dl = 1, (1 if bi == 0 else 2)
            t32 = _ACC["fp32_trunk_from"]
            y = bottleneck(y, P, f"{pfx}layer{li + 1}.{bi}.", stride, q, trunk_round=not (t32 and li + 1 >= t32), dilation=dl)
        outs.append(y)
    return outs


def mask_downsample(mask, size):
    """backbone.py:107: F.interpolate(m[None].float(), size) (nearest): src = floor(dst * in / out)."""
    H, W = mask.shape[-2:]
    h, w = size
    iy = torch.floor(torch.arange(h, dtype=torch.float32) * (H / h)).long().clamp(max=H - 1)
    ix = torch.floor(torch.arange(w, dtype=torch.float32) * (W / w)).long().clamp(max=W - 1)
    return mask[:, iy][:, :, ix]


def learned_pos(P, B, h, w, pfx="img_backbone.1."):
    """PositionEmbeddingLearned.forward (models/modeling/position_encoding.py:74-84): [col_embed[x] | row_embed[y]] per pixel."""
    x_emb = P[pfx + "col_embed.weight"][:w]
    y_emb = P[pfx + "row_embed.weight"][:h]
    pos = torch.cat([x_emb.unsqueeze(0).repeat(h, 1, 1), y_emb.unsqueeze(1).repeat(1, w, 1)], dim=-1)
    return pos.permute(2, 0, 1).unsqueeze(0).repeat(B, 1, 1, 1)       # [B, 256, h, w]


def sine_pos(mask, num_pos_feats=128, temperature=10000.0):
    """models/modeling/position_encoding.py:36-56 (normalize=True, scale=2*pi, eps=1e-6)."""
    not_mask = ~mask
    y_embed = not_mask.cumsum(1, dtype=torch.float32)
    x_embed = not_mask.cumsum(2, dtype=torch.float32)
    eps, scale = 1e-6, 2 * math.pi
    y_embed = (y_embed - 0.5) / (y_embed[:, -1:, :] + eps) * scale
    x_embed = (x_embed - 0.5) / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / num_pos_feats)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    pos_x = torch.stack((pos_x[..., 0::2].sin(), pos_x[..., 1::2].cos()), dim=4).flatten(3)
    pos_y = torch.stack((pos_y[..., 0::2].sin(), pos_y[..., 1::2].cos()), dim=4).flatten(3)
    return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)   # [B, 256, h, w]


# ----------------------------------------------------------------------------------------------
# small building blocks
# ----------------------------------------------------------------------------------------------
def linear(x, P, pfx, q=False):
    return linear_acc(rq(x, q), rq_fwd(P[pfx + "weight"], q), P[pfx + "bias"])


def layer_norm(x, P, pfx, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), P[pfx + "weight"], P[pfx + "bias"], eps)


def drop(x, p, train):
    return F.dropout(x, p, train) if (train and p > 0) else x


def mlp_mapping(x, P, pfx, q=False, train=False):
    """models/reftr_transformer.py:14-23: Linear-LN-ReLU-Dropout(0.1)-Linear-LN-ReLU."""
    y = F.relu(layer_norm(linear(x, P, pfx + "0.", q), P, pfx + "1."))
    y = drop(y, 0.1, train)
    return F.relu(layer_norm(linear(y, P, pfx + "4.", q), P, pfx + "5."))


def mha(P, pfx, query, key, value, key_padding_mask, nheads, p_drop=0.0, train=False, q=False):
    """nn.MultiheadAttention as the reference uses it (models/modeling/transformer.py:151,174-175,
    211-212,239-246): packed in_proj [3E,E], q scaled by 1/sqrt(dh), key_padding_mask -> -inf,
    softmax, dropout on the probabilities, out_proj.  Inputs are [L, B, E] (seq-first)."""
    E = query.shape[-1]
    dh = E // nheads
    W, bias = P[pfx + "in_proj_weight"], P[pfx + "in_proj_bias"]
    Lq, B, _ = query.shape
    Lk = key.shape[0]
    qh = linear_acc(rq(query, q), rq_fwd(W[:E], q), bias[:E]) * (dh ** -0.5)
    kh = linear_acc(rq(key, q), rq_fwd(W[E:2 * E], q), bias[E:2 * E])
    vh = linear_acc(rq(value, q), rq_fwd(W[2 * E:], q), bias[2 * E:])
    qh = rq(qh, q).reshape(Lq, B * nheads, dh).transpose(0, 1)
    kh = rq(kh, q).reshape(Lk, B * nheads, dh).transpose(0, 1)
    vh = rq(vh, q).reshape(Lk, B * nheads, dh).transpose(0, 1)
    scores = bmm_acc(qh, kh.transpose(1, 2))
    if key_padding_mask is not None:
        scores = scores.view(B, nheads, Lq, Lk).masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
        scores = scores.view(B * nheads, Lq, Lk)
    attn = drop(F.softmax(scores, dim=-1), p_drop, train)
    # q-mode: the MFMA attention kernels (rt_attn_fwd / rt_attn_bwd) hand the probabilities to the P V product as bf16 operands;
    # the one-query kernels (a single query row per image: the single-phrase decoder) keep them in fp32 (csrc/rt_attention.hip)
    attn = rq(attn, q and Lq > 1)
    out = bmm_acc(attn, vh).transpose(0, 1).reshape(Lq, B, E)
    return linear_acc(rq(out, q), rq_fwd(P[pfx + "out_proj.weight"], q), P[pfx + "out_proj.bias"])


def encoder_layer(x, pos, kpm, P, pfx, cfg, train=False, q=False):
    """TransformerEncoderLayer.forward_post, models/modeling/transformer.py:168-181."""
    qk = x + pos
    a = mha(P, pfx + "self_attn.", qk, qk, x, kpm, cfg.nheads, cfg.dropout, train, q)
    x = layer_norm(x + drop(a, cfg.dropout, train), P, pfx + "norm1.")
    h = drop(F.relu(linear(x, P, pfx + "linear1.", q)), cfg.dropout, train)
    f = linear(h, P, pfx + "linear2.", q)
    return layer_norm(x + drop(f, cfg.dropout, train), P, pfx + "norm2.")


def decoder_layer(t, memory, qpos, pos, tgt_kpm, mem_kpm, P, pfx, cfg, train=False, q=False):
    """TransformerDecoderLayer.forward_post, models/modeling/transformer.py:231-252."""
    qk = t + qpos
    a = mha(P, pfx + "self_attn.", qk, qk, t, tgt_kpm, cfg.nheads, cfg.dropout, train, q)
    t = layer_norm(t + drop(a, cfg.dropout, train), P, pfx + "norm1.")
    a = mha(P, pfx + "multihead_attn.", t + qpos, memory + pos, memory, mem_kpm, cfg.nheads, cfg.dropout, train, q)
    t = layer_norm(t + drop(a, cfg.dropout, train), P, pfx + "norm2.")
    h = drop(F.relu(linear(t, P, pfx + "linear1.", q)), cfg.dropout, train)
    f = linear(h, P, pfx + "linear2.", q)
    return layer_norm(t + drop(f, cfg.dropout, train), P, pfx + "norm3.")


# ----------------------------------------------------------------------------------------------
# BERT (HF BertModel, post-LN, erf-GELU, eps 1e-12, tanh pooler) — SURVEY.md A5
# ----------------------------------------------------------------------------------------------
def bert_forward(P, ids, attn_mask, bc: BertCfg, pfx="lang_backbone.", train=False, q=False):
    B, L = ids.shape
    e = pfx + "embeddings."
    if bc.pad_idx >= 0:       # HF RobertaEmbeddings: positions count the non-pad tokens, offset by the padding index
        tok = ids.ne(bc.pad_idx).int()
        pos_ids = (torch.cumsum(tok, dim=1) * tok).long() + bc.pad_idx
        pe = P[e + "position_embeddings.weight"][pos_ids]
    else:
        pe = P[e + "position_embeddings.weight"][:L][None]
    h = P[e + "word_embeddings.weight"][ids] + pe + P[e + "token_type_embeddings.weight"][0][None, None]
    h = drop(layer_norm(h, P, e + "LayerNorm.", bc.eps), bc.dropout, train)
    dh = bc.hidden // bc.heads
    # HF extended mask: (1 - mask) * finfo.min added to the scores
    add_mask = (1.0 - attn_mask.to(torch.float32))[:, None, None, :] * torch.finfo(torch.float32).min
    for i in range(bc.layers):
        lp = f"{pfx}encoder.layer.{i}."
        qh = rq(linear(h, P, lp + "attention.self.query.", q), q).view(B, L, bc.heads, dh).transpose(1, 2)
        kh = rq(linear(h, P, lp + "attention.self.key.", q), q).view(B, L, bc.heads, dh).transpose(1, 2)
        vh = rq(linear(h, P, lp + "attention.self.value.", q), q).view(B, L, bc.heads, dh).transpose(1, 2)
        s = matmul_acc(qh, kh.transpose(-1, -2)) / math.sqrt(dh) + add_mask
        a = rq(drop(F.softmax(s, dim=-1), bc.dropout, train), q)          # bf16 operand of the P V product (rt_attn_fwd)
        ctx = matmul_acc(a, vh).transpose(1, 2).reshape(B, L, bc.hidden)
        o = drop(linear(ctx, P, lp + "attention.output.dense.", q), bc.dropout, train)
        h = layer_norm(h + o, P, lp + "attention.output.LayerNorm.", bc.eps)
        f = F.gelu(linear(h, P, lp + "intermediate.dense.", q))
        f = drop(linear(f, P, lp + "output.dense.", q), bc.dropout, train)
        h = layer_norm(h + f, P, lp + "output.LayerNorm.", bc.eps)
    pooled = torch.tanh(linear(h[:, 0], P, pfx + "pooler.dense.", q))
    return h, pooled


# ----------------------------------------------------------------------------------------------
# RefTR model (models/reftr_transformer.py:159-297, models/reftr.py:51-120)
# ----------------------------------------------------------------------------------------------
def query_encoder(P, ctx, phrase_feat, mask_ctx, cfg, pfx="query_encoder.", train=False, q=False):
    """QueryEncoder.forward, models/reftr_transformer.py:41-66.  ctx [B,L,E], phrase_feat [B,n_ph,E],
    mask_ctx [B,n_ph,L] (True = ignore).  NOTE: attention logits are NOT scaled by 1/sqrt(d) (:51)."""
    B, n_ph, E = phrase_feat.shape
    k = linear(ctx[:, 0:1, :], P, pfx + "linear1.", q)
    qs = linear(ctx, P, pfx + "linear2.", q).transpose(1, 2)
    v = linear(ctx, P, pfx + "linear3.", q).unsqueeze(1)
    w = bmm_acc(k, qs).expand(-1, n_ph, -1).masked_fill(mask_ctx, float("-inf"))
    w = F.softmax(w, dim=-1).unsqueeze(-1)
    c = (v * w).sum(dim=-2)
    c = layer_norm(linear(c, P, pfx + "context_out.0.", q), P, pfx + "context_out.1.")
    c = ctx[:, None, 0, :] + c
    f = mlp_mapping(torch.cat([c, phrase_feat], dim=-1), P, pfx + "fuse_encoder_query.", q, train)
    emb = P[pfx + "query_embed.weight"]                       # [n_q, 2E]
    n_q = emb.shape[0]
    pq = f.view(B, n_ph, 1, -1).repeat(1, 1, 1, 2) + emb.view(1, 1, n_q, -1)
    pq = pq.view(B, n_ph * n_q, -1).transpose(0, 1)           # [n_ph*n_q, B, 2E]
    return pq[..., :E], pq[..., E:]


def context_masks(samples):
    """models/reftr_transformer.py:206-248 — integer/bool only.  Returns (mask_context [B,n_ph,L] True=ignore,
    query_mask [B, n_ph] True=ignore (before the n_q expand))."""
    sm = samples["sentence_mask"].to(torch.bool)
    B, L = sm.shape
    if "phrase" in samples:
        pl, pr = samples["phrase_pos_l"], samples["phrase_pos_r"]
        ar = torch.arange(L)[None, None, :]
        mask_context = ~((ar >= pl[..., None]) & (ar < pr[..., None]))
        query_mask = ~samples["phrase_mask"].to(torch.bool)[:, :, 2]
    else:
        slen = sm.to(torch.int32).sum(-1)
        mask_context = (~sm).view(B, 1, L).clone()
        mask_context[:, :, 0] = True
        mask_context[torch.arange(B), 0, slen - 1] = True
        query_mask = torch.zeros((B, 1), dtype=torch.bool)
    return mask_context, query_mask


def reftr_forward(P, samples, cfg: Cfg, train=False, q=False):
    """RefTR.forward.  samples = {'img' [B,3,H,W] f32, 'img_mask' [B,H,W] bool (True = pad),
    'sentence' int64 [B,L], 'sentence_mask' [B,L], optional phrase fields}.  Returns a dict with the
    reference outputs plus intermediates used by the parity tests."""
    img, img_mask = samples["img"], samples["img_mask"]
    B = img.shape[0]
    E = cfg.hidden
    feats = resnet_body(img, P, layers=cfg.resnet_layers, q=q, dilation=cfg.dilation)
    c5 = feats[-1]
    m5 = mask_downsample(img_mask, c5.shape[-2:])
    pos5 = learned_pos(P, m5.shape[0], m5.shape[1], m5.shape[2]) if cfg.pos_learned else sine_pos(m5, E // 2)
    # input_proj: 1x1 conv + GroupNorm(32) (models/reftr_transformer.py:121-125,174)
    src = conv2d_acc(rq(c5, q), rq_fwd(P["input_proj.0.0.weight"], q), P["input_proj.0.0.bias"])
    src = F.group_norm(src, 32, P["input_proj.0.1.weight"], P["input_proj.0.1.bias"], 1e-5)

    sent, smask = samples["sentence"], samples["sentence_mask"]
    L = sent.shape[1]
    seq, pooled = bert_forward(P, sent, smask, cfg.bert, train=train, q=q)
    sent_feat = mlp_mapping(seq, P, "map_sentence.", q, train)                     # :201
    n_q = cfg.n_q
    mask_context, query_mask = context_masks(samples)
    if "phrase" in samples:
        ph = samples["phrase"]
        n_ph = ph.shape[1]
        _, ph_pooled = bert_forward(P, ph.view(B * n_ph, -1), samples["phrase_mask"].view(B * n_ph, -1),
                                    cfg.bert, train=train, q=q)                     # :217
    else:
        n_ph = 1
        ph_pooled = pooled
    query_mask = query_mask[:, :, None].expand(-1, -1, n_q).reshape(B, n_ph * n_q)
    ph_feat = mlp_mapping(ph_pooled, P, "map_phrase.", q, train).view(B, n_ph, -1)   # :250

    # VLTransformer.encode (models/reftr.py:51-120): [lang; img] along S, seq-first layout
    vt = "vl_transformer."
    img_src = src.flatten(2).permute(2, 0, 1)                                       # [HW, B, E]
    img_pos = pos5.flatten(2).permute(2, 0, 1) + P[vt + "level_embed"][0].view(1, 1, -1) \
        + P[vt + "token_type_embeddings.weight"][1].view(1, 1, -1)
    lang_src = sent_feat.transpose(0, 1)                                            # [L, B, E]
    lang_pos = (P[vt + "lang_pos_embeddings.weight"][:L] + P[vt + "token_type_embeddings.weight"][0][None])
    lang_pos = lang_pos[:, None, :].expand(-1, B, -1)
    kpm = torch.cat([~smask.to(torch.bool), m5.flatten(1)], dim=1)                  # [B, S] True = ignore
    x = torch.cat([lang_src, img_src], dim=0)
    pos = torch.cat([lang_pos, img_pos], dim=0)
    for i in range(cfg.enc_layers):
        x = encoder_layer(x, pos, kpm, P, f"{vt}encoder.layers.{i}.", cfg, train, q)
    memory = x

    tgt, qpos = query_encoder(P, memory[:L].transpose(0, 1), ph_feat, mask_context, cfg, train=train, q=q)
    hs = []
    t = tgt
    for i in range(cfg.dec_layers):
        t = decoder_layer(t, memory, qpos, pos, query_mask, kpm, P, f"{vt}decoder.layers.{i}.", cfg, train, q)
        hs.append(layer_norm(t, P, vt + "decoder.norm."))                           # transformer.py:131-138
    hs = torch.stack(hs).transpose(1, 2)                                            # [nl, B, n_ph*n_q, E]
    hs = hs.reshape(len(hs), B, n_ph, n_q, -1)
    y = F.relu(linear(hs, P, "bbox_embed.layers.0.", q))
    y = F.relu(linear(y, P, "bbox_embed.layers.1.", q))
    logits = linear(y, P, "bbox_embed.layers.2.", q)                                 # backbone.py:26-38
    boxes = logits.sigmoid()
    phrase_mask = ~query_mask
    out = {"pred_boxes": boxes[-1], "phrase_mask": phrase_mask, "logits": logits, "memory": memory,
           "c5": c5, "src": src, "pos5": pos5, "kpm": kpm, "feats": feats, "hs": hs}
    if cfg.aux_loss:
        out["aux_outputs"] = [{"pred_boxes": b, "phrase_mask": phrase_mask} for b in boxes[:-1]]
    if cfg.masks:
        out.update(seg_forward(P, out, samples, cfg, q=q))
    return out


# ----------------------------------------------------------------------------------------------
# RES head (models/reftr_segmentation.py:151-280): MHAttentionMap + MaskHeadSmallConv
# ----------------------------------------------------------------------------------------------
def mh_attention_map(P, qv, k, mask, nheads, pfx="bbox_attention.", q=False):
    """MHAttentionMap.forward (reftr_segmentation.py:196-208): softmax over (heads, h, w) JOINTLY; dropout 0.
    qv [B,Q,E], k [B,E,h,w], mask [B,h,w] bool (True = pad) -> [B,Q,nheads,h,w]."""
    E = qv.shape[-1]
    qq = linear(qv, P, pfx + "q_linear.", q)
    kk = conv2d_acc(rq(k, q), rq_fwd(P[pfx + "k_linear.weight"], q)[:, :, None, None], P[pfx + "k_linear.bias"])
    qh = qq.view(qq.shape[0], qq.shape[1], nheads, E // nheads)
    kh = kk.view(kk.shape[0], nheads, E // nheads, kk.shape[-2], kk.shape[-1])
    w = torch.einsum("bqnc,bnchw->bqnhw", qh * float(E / nheads) ** -0.5, kh)
    w = w.masked_fill(mask[:, None, None], float("-inf"))
    return F.softmax(w.flatten(2), dim=-1).view_as(w)


MASK_HEAD_TRACE = None      # debugging aid (benchmarks/debug_seg_floor.py): a dict that receives the head's intermediate tensors


def mask_head(P, x, bbox_mask, fpns, pfx="mask_head.", q=False):
    """MaskHeadSmallConv.forward (reftr_segmentation.py:240-280).  x [B,2E,h,w], bbox_mask [B,Q,n,h,w],
    fpns = [stride16, stride8, stride4 features].  Returns (mask logits [B*Q,1,4h',4w'..], last feature)."""
    Q = bbox_mask.shape[1]

    def expand(t, n):
        return t.unsqueeze(1).repeat(1, int(n), 1, 1, 1).flatten(0, 1)

    def conv(t, name, pad):
        return conv2d_acc(rq(t, q), rq_fwd(P[pfx + name + ".weight"], q), P[pfx + name + ".bias"], padding=pad)

    def gn(t, name):
        return F.relu(F.group_norm(t, 8, P[pfx + name + ".weight"], P[pfx + name + ".bias"], 1e-5))

    tr = MASK_HEAD_TRACE if MASK_HEAD_TRACE is not None else {}
    x = torch.cat([expand(x, Q), bbox_mask.flatten(0, 1)], 1)
    tr["X0"] = x
    tr["u1"] = conv(x, "lay1", 1); x = gn(tr["u1"], "gn1"); tr["a1"] = x
    tr["u2"] = conv(x, "lay2", 1); x = gn(tr["u2"], "gn2"); tr["a2"] = x
    for i, f in enumerate(fpns):
        cur = conv(f, f"adapter{i + 1}", 0)
        tr[f"fo{i}"] = cur
        if cur.shape[0] != x.shape[0]:
            cur = expand(cur, x.shape[0] // cur.shape[0])
        # q-mode: GroupNorm + ReLU leaves its output as a bf16 operand in the HIP path (rt_gn_nhwc_fwd), also where the next consumer
        # is the FPN's upsample + add (rt_upsample_add) rather than a convolution
        x = cur + F.interpolate(rq(x, q), size=cur.shape[-2:], mode="nearest")
        tr[f"x{i}"] = x
        tr[f"u{i + 3}"] = conv(x, f"lay{i + 3}", 1)
        x = gn(tr[f"u{i + 3}"], f"gn{i + 3}"); tr[f"a{i + 3}"] = x
    return conv(x, "out_lay", 1), x


def seg_forward(P, out, samples, cfg: Cfg, q=False):
    """RefTRSeg.forward's RES part (reftr_segmentation.py:136-176): last decoder layer only, n_ph = 1."""
    L = samples["sentence"].shape[1]
    B = samples["img"].shape[0]
    src = out["src"]
    h, w = src.shape[-2:]
    mem_vis = out["memory"][L:].permute(1, 2, 0).reshape(B, cfg.hidden, h, w)
    m5 = mask_downsample(samples["img_mask"], (h, w))
    hs_last = out["hs"][-1]                                   # [B, n_ph, n_q, E]
    bbox_mask = mh_attention_map(P, hs_last.reshape(B, -1, cfg.hidden), mem_vis, m5, cfg.nheads, q=q)
    feats = out["feats"]
    seg, res_feat = mask_head(P, torch.cat([src, mem_vis], 1), bbox_mask, [feats[2], feats[1], feats[0]], q=q)
    res = {"pred_masks": seg.view(B, -1, seg.shape[-2], seg.shape[-1]), "mask_att": bbox_mask[:, 0], "res_feat": res_feat}
    if cfg.cem:
        res["cem_loss"] = cem_forward(P, hs_last, res_feat, q=q)
    return res


def cem_forward(P, rec_feat, res_feat, pfx="cem_block.", q=False):
    """CEM.forward (reftr_segmentation.py:25-41).  rec_feat [B, n_ph, n_q, c] (last decoder layer), res_feat [B, c/16, H, W]
    (MaskHeadSmallConv's last feature map).  es = softmax over the n_ph*n_q axis (= 1 for the single query of RES); ec =
    softmax over the pixels of c2(res); tsc = clamp((cos(c3(rec), res) + 1) / 2, 1e-6, 1 - 1e-6);
    loss = -sum_b log(es^T tsc ec + 1e-6) / B."""
    B, n_ph, n_q, c = rec_feat.shape
    rec = rec_feat.reshape(B, -1, c)
    res = res_feat.reshape(B, c // 16, -1).transpose(1, 2)
    es = F.softmax(linear(rec, P, pfx + "c1.", q), dim=-2)
    ec = F.softmax(linear(res, P, pfx + "c2.", q=False), dim=-2)
    r = F.normalize(linear(rec, P, pfx + "c3.", q), dim=-1)
    s = F.normalize(res, dim=-1).transpose(-1, -2)
    tsc = torch.clamp((bmm_acc(r, s) + 1.0) / 2.0, 1e-6, 1.0 - 1e-6)
    energy = bmm_acc(bmm_acc(es.transpose(-1, -2), tsc), ec)
    return -1.0 * torch.sum(torch.log(energy + 1e-6)) * 1.0 / B


def dice_loss(inputs, targets, num_boxes):
    """models/modeling/segmentation.py:178-194."""
    inputs = inputs.sigmoid().flatten(1)
    num = 2 * (inputs * targets).sum(1)
    den = inputs.sum(-1) + targets.sum(-1)
    return (1 - (num + 1) / (den + 1)).sum() / num_boxes


def sigmoid_focal_loss(inputs, targets, num_boxes, alpha=0.25, gamma=2.0):
    """models/modeling/segmentation.py:197-221."""
    prob = inputs.sigmoid()
    ce = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = prob * targets + (1 - prob) * (1 - targets)
    loss = ce * ((1 - p_t) ** gamma)
    loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    return loss.mean(1).sum() / num_boxes


def pad_masks(targets):
    """nested_tensor_from_tensor_list over t['masks'] ([1,h,w] bool): zero-pad to the batch maximum (util/misc.py:288-305)."""
    hs = max(t["masks"].shape[-2] for t in targets); ws = max(t["masks"].shape[-1] for t in targets)
    out = torch.zeros(len(targets), targets[0]["masks"].shape[0], hs, ws)
    for i, t in enumerate(targets):
        m = t["masks"]
        out[i, :, :m.shape[-2], :m.shape[-1]] = m.float()
    return out


def loss_masks(pred_masks, targets):
    """CriterionVGOnePhraseSeg.loss_masks (reftr_segmentation.py:314-337): bilinear upsample (align_corners=False) to the
    padded target size; both losses normalised by bs * num_q."""
    bs, nq = pred_masks.shape[:2]
    tm = pad_masks(targets).to(pred_masks)
    src = F.interpolate(pred_masks, size=tm.shape[-2:], mode="bilinear", align_corners=False)
    src = src.view(bs * nq, -1); tm = tm.view(bs * nq, -1)
    return {"loss_mask": sigmoid_focal_loss(src, tm, bs * nq), "loss_dice": dice_loss(src, tm, bs * nq)}


# ----------------------------------------------------------------------------------------------
# criterion (models/criterion.py:113-202, util/box_ops.py:17-69)
# ----------------------------------------------------------------------------------------------
def cxcywh_to_xyxy(b):
    cx, cy, w, h = b.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def giou_diag(a, b):
    """diag(generalized_box_iou(a, b)) (util/box_ops.py:32-69) for xyxy boxes a[i], b[i]."""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(a[:, :2], b[:, :2]); rb = torch.min(a[:, 2:], b[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, 0] * wh[:, 1]
    union = area_a + area_b - inter
    iou = inter / union
    lt2 = torch.min(a[:, :2], b[:, :2]); rb2 = torch.max(a[:, 2:], b[:, 2:])
    wh2 = (rb2 - lt2).clamp(min=0)
    area = wh2[:, 0] * wh2[:, 1]
    return iou - (area - union) / area


def loss_boxes(pred_boxes, phrase_mask, targets, num_boxes):
    """CriterionVGMultiPhrase.loss_boxes, models/criterion.py:113-153.  pred_boxes [B,n_ph,k,4],
    phrase_mask [B, n_ph*k] (True = valid), targets list of {'boxes': [n,4]}."""
    B, n_ph, k, _ = pred_boxes.shape
    m = phrase_mask.view(B, n_ph, k)
    preds, tgts = [], []
    for i in range(B):
        pi = pred_boxes[i][m[i]].view(-1, k, 4)       # masked_select keeps phrase order
        assert pi.shape[0] == targets[i]["boxes"].shape[0]
        preds.append(pi); tgts.append(targets[i]["boxes"])
    p = torch.cat(preds, 0)
    t = torch.cat(tgts, 0).unsqueeze(1).expand(-1, k, -1)
    p = p.reshape(-1, 4); t = t.reshape(-1, 4)
    l1 = (p - t).abs().sum() / (num_boxes * k)
    gi = (1 - giou_diag(cxcywh_to_xyxy(p), cxcywh_to_xyxy(t))).sum() / (num_boxes * k)
    return {"loss_bbox": l1, "loss_giou": gi}


def criterion(out, targets, world_size=1, global_num_boxes=None):
    """CriterionVGMultiPhrase.forward, models/criterion.py:166-202 (num_boxes all-reduce modelled by
    `global_num_boxes` / world_size)."""
    nb = sum(len(t["labels"]) for t in targets) if global_num_boxes is None else global_num_boxes
    nb = max(nb / world_size, 1.0)
    losses = dict(loss_boxes(out["pred_boxes"], out["phrase_mask"], targets, nb))
    if "pred_masks" in out:       # CriterionVGOnePhraseSeg: losses = ['masks', 'boxes'] (reftr_segmentation.py:388)
        losses.update(loss_masks(out["pred_masks"], targets))
        if "cem_loss" in out:     # reftr_segmentation.py:335-336
            losses["loss_cem"] = out["cem_loss"]
    for i, aux in enumerate(out.get("aux_outputs", [])):
        for k_, v in loss_boxes(aux["pred_boxes"], aux["phrase_mask"], targets, nb).items():
            losses[f"{k_}_{i}"] = v
    return losses


# ----------------------------------------------------------------------------------------------
# evaluation post-processing (models/post_process.py:45-83, models/reftr_segmentation.py:282-302)
# ----------------------------------------------------------------------------------------------
def postprocess_boxes(pred_boxes, phrase_mask, target_sizes, scale_to_original_shape=False):
    """PostProcessVGMultiPhrase.forward: per image, the valid phrases' first prediction as xyxy (phrase order kept),
    optionally times [img_w, img_h, img_w, img_h] (target_sizes [B,2] = (h, w), integer -> promoted to fp32)."""
    B, n_ph, k, _ = pred_boxes.shape
    m = phrase_mask.view(B, n_ph, k, -1)
    out = []
    for i in range(B):
        pi = torch.masked_select(pred_boxes[i], m[i]).view(-1, k, 4)
        b = cxcywh_to_xyxy(pi[:, 0, :])
        if scale_to_original_shape:
            ih, iw = target_sizes[i:i + 1].unbind(1)
            b = b * torch.stack([iw, ih, iw, ih], dim=1)
        out.append(b)
    return out


def postprocess_segm(pred_masks, orig_target_sizes, max_target_sizes, threshold=0.5):
    """PostProcessSegm.forward: bilinear (align_corners=False) to the padded frame, sigmoid > threshold, crop to each image's
    own size, nearest resize to the original size.  Returns [(masks bool [Q,1,h_i,w_i], masks_origin uint8 [Q,1,H_i,W_i])]."""
    max_h, max_w = max_target_sizes.max(0)[0].tolist()
    m = F.interpolate(pred_masks.squeeze(2), size=(max_h, max_w), mode="bilinear", align_corners=False)
    m = m.sigmoid() > threshold
    out = []
    for cur, t, tt in zip(m, max_target_sizes, orig_target_sizes):
        c = cur[:, :int(t[0]), :int(t[1])].unsqueeze(1)
        out.append((c, F.interpolate(c.float(), size=tuple(tt.tolist()), mode="nearest").byte()))
    return out


def weight_dict(cfg: Cfg):
    """models/reftr_transformer.py:320-329."""
    wd = {"loss_giou": cfg.giou_loss_coef, "loss_bbox": cfg.bbox_loss_coef}
    if cfg.masks:                 # build_reftr_seg, reftr_segmentation.py:349-351
        wd.update({"loss_dice": cfg.dice_loss_coef, "loss_mask": cfg.mask_loss_coef, "loss_cem": 1.0})
    if cfg.aux_loss:
        aux = {}
        for i in range(cfg.dec_layers - 1):
            aux.update({f"{k}_{i}": v for k, v in wd.items()})
        aux.update({k + "_enc": v for k, v in wd.items()})
        wd.update(aux)
    return wd


def total_loss(losses, wd):
    """engine_vg.py:43."""
    return sum(losses[k] * wd[k] for k in losses if k in wd)


# ----------------------------------------------------------------------------------------------
# training step (engine_vg.py:40-72, main_vg.py:234-270)
# ----------------------------------------------------------------------------------------------
def is_trainable(name):
    """backbone.py:87-89: conv1 / layer1 frozen; FrozenBN tensors are buffers."""
    if name.startswith("img_backbone.1."):          # PositionEmbeddingLearned tables
        return True
    if name.startswith("img_backbone."):
        if any(s in name for s in ("running_mean", "running_var", ".bn", "downsample.1.")):
            return False
        return any(s in name for s in ("layer2", "layer3", "layer4"))
    return True


def lr_group(name, lr=1e-4, lr_backbone=1e-5, lr_bert=1e-5):
    """main_vg.py:29-33,234-262: backbone names -> lr_backbone, lang_backbone -> lr_bert."""
    if "img_backbone.0" in name:
        return lr_backbone
    if "lang_backbone" in name:
        return lr_bert
    return lr


def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ (engine_vg.py:63): total = ||(||g_i||)||, coef = min(1, max/(total+1e-6))."""
    total = torch.sqrt(sum((g.detach().float() ** 2).sum() for g in grads.values()))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return total, {k: g * coef for k, g in grads.items()}


def adamw_step(P, grads, state, step, lrs, betas=(0.9, 0.999), eps=1e-8, wd=1e-4):
    """torch.optim.AdamW, decoupled weight decay (main_vg.py:264-268); SURVEY.md A16."""
    b1, b2 = betas
    for k, g in grads.items():
        p = P[k]
        st = state.setdefault(k, {"m": torch.zeros_like(p), "v": torch.zeros_like(p)})
        lr = lrs[k]
        p.mul_(1 - lr * wd)
        st["m"].mul_(b1).add_(g, alpha=1 - b1)
        st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** step
        bc2 = 1 - b2 ** step
        denom = (st["v"].sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(st["m"], denom, value=-lr / bc1)


def train_step(P, samples, targets, cfg, state, step, max_norm=0.1, train=True, q=False,
               lr=1e-4, lr_backbone=1e-5, lr_bert=1e-5):
    """One iteration of engine_vg.train_one_epoch's loop body (engine_vg.py:40-72).  P is updated in place;
    returns (losses dict, total weighted loss, grad norm, grads)."""
    names = [k for k in P if is_trainable(k) and torch.is_floating_point(P[k])]
    leaves = {k: P[k].detach().requires_grad_(True) for k in names}
    Pl = dict(P); Pl.update(leaves)
    out = reftr_forward(Pl, samples, cfg, train=train, q=q)
    losses = criterion(out, targets)
    loss = total_loss(losses, weight_dict(cfg))
    gl = torch.autograd.grad(loss, [leaves[k] for k in names], allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(P[k])) for k, g in zip(names, gl)}
    gnorm, grads_c = clip_grad_norm(grads, max_norm) if max_norm > 0 else (None, grads)
    lrs = {k: lr_group(k, lr, lr_backbone, lr_bert) for k in names}
    with torch.no_grad():
        adamw_step(P, grads_c, state, step, lrs)
    return {k: float(v) for k, v in losses.items()}, float(loss), (float(gnorm) if gnorm is not None else None), grads
