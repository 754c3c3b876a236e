"""Forward and hand-written backward of the language / visual-language side of RefTR on the HIP kernels:
HF-style BERT encoder, mlp_mapping, VLTransformer encoder + decoder, QueryEncoder, bbox head.

Reference arithmetic (all fp32 there): models/reftr_transformer.py:14-66,159-297, models/reftr.py:51-120,
models/modeling/transformer.py:81-143,168-181,231-252, models/modeling/backbone.py:26-38, HF BertModel
(SURVEY.md Appendix A5-A12).  Here: bf16 GEMM operands / fp32 accumulation on MFMA, fp32 residual stream,
fp32 LayerNorm / softmax statistics.  Token layout is BATCH-major ([B*S, E]; the reference is [S, B, E]),
language tokens first inside every image's S = L + h*w rows (models/reftr.py:115-117).

Backward is explicit: each `*_bwd` consumes the context its `*_fwd` returned; weight gradients are
accumulated by the kernels straight into the flat gradient buffer (ParamStore.G).
"""
import math
import os

import torch

from .. import hip as H

RELU, GELU, TANH, NONE = H.ACT_RELU, H.ACT_GELU, H.ACT_TANH, H.ACT_NONE


class Lin:
    """One nn.Linear as the kernels see it: bf16 operand W [N,K] (forward), WT [K,N] (backward-data),
    fp32 master / gradient views for the weight-prep and weight-gradient kernels."""
    __slots__ = ("w32", "b32", "gw", "gb", "W", "WT", "N", "K")

    def __init__(self, w32, b32, gw, gb, device):
        self.w32, self.b32, self.gw, self.gb = w32, b32, gw, gb
        self.N, self.K = w32.shape
        self.W = torch.empty(self.N, self.K, dtype=torch.bfloat16, device=device)
        self.WT = torch.empty(self.K, self.N, dtype=torch.bfloat16, device=device)

    def refresh(self):
        H.weight_prep(self.w32, self.N, 1, self.K, dst=self.W, dst_t=self.WT)


class Net:
    def __init__(self, store, cfg):
        self.store, self.cfg = store, cfg
        self.lins = {}
        self.training = True
        self._seed_base = 0x1234567
        self._site = 0
        import os
        # bit 0: BERT branch on its own stream; bit 1: weight-gradient launches on their own stream
        on = int(os.environ.get("REFTR_STREAMS", "1")) if str(store.device).startswith("cuda") else 0
        self.wg = H.SideStream(bool(on & 2), defer=True)
        self.side = H.SideStream(bool(on & 1))
        self.small_wg = H.SmallWgradBatch()
        self.trivial_sa = True            # one-key self-attention (one query per image): softmax == 1, the attention launch is skipped
        self.fold_sa = os.environ.get("REFTR_FOLD_SA", "1") != "0"
        # the decoder's forward chain as one cooperative launch (csrc/rt_decoder.hip) whenever its shape allows
        self.dec_coop = os.environ.get("REFTR_DEC_COOP", "1") != "0"
        self.dec_coop_bwd = os.environ.get("REFTR_DEC_COOP_BWD", "1") != "0"
        self.dec_kv_pack = os.environ.get("REFTR_DEC_KV_PACK", "1") != "0"       # d memory as one K-concatenated product
        self._kv_cat = None
        self._dec_cus = None
        # allocated (and zeroed) NOW, outside any stream capture: a zero-fill captured into a graph would reset the launch epoch on
        # every replay and make the previous replay's hand-off tags look current
        self._dec_handoff = H.decoder_handoff(store.device) if str(store.device).startswith("cuda") else None
        self.ln_batch = H.LnGradBatch() if str(store.device).startswith("cuda") else None
        self.big_wg = H.WgradBatch() if str(store.device).startswith("cuda") else None
        self._build_lins()

    # ------------------------------------------------------------------ operand bank
    def _lin(self, key, wname, bname, rows=None):
        P, G = self.store.P, self.store.G
        w, gw, b, gb = P[wname], G[wname], P[bname], G[bname]
        if rows is not None:
            a, e = rows
            w, gw, b, gb = w[a:e], gw[a:e], b[a:e], gb[a:e]
        self.lins[key] = Lin(w, b, gw, gb, self.store.device)

    def _build_lins(self):
        cfg, st = self.cfg, self.store
        E = cfg.hidden
        for i in range(cfg.bert.layers):
            lp = f"lang_backbone.encoder.layer.{i}."
            q = lp + "attention.self.query."
            self.lins[lp + "qkv"] = Lin(st.packed(q + "weight", 3), st.packed(q + "bias", 3),
                                        st.packed(q + "weight", 3, grad=True), st.packed(q + "bias", 3, grad=True), st.device)
            for n in ("attention.output.dense.", "intermediate.dense.", "output.dense."):
                self._lin(lp + n, lp + n + "weight", lp + n + "bias")
        self._lin("lang_backbone.pooler.dense.", "lang_backbone.pooler.dense.weight", "lang_backbone.pooler.dense.bias")
        for m in ("map_sentence.", "map_phrase.", "query_encoder.fuse_encoder_query."):
            for j in ("0.", "4."):
                self._lin(m + j, m + j + "weight", m + j + "bias")
        for n in ("linear1.", "linear2.", "linear3.", "context_out.0."):
            self._lin("query_encoder." + n, "query_encoder." + n + "weight", "query_encoder." + n + "bias")
        vt = "vl_transformer."
        for i in range(cfg.enc_layers):
            p = f"{vt}encoder.layers.{i}."
            self._mha(p + "self_attn.", split_qk=False)
            self._lin(p + "linear1.", p + "linear1.weight", p + "linear1.bias")
            self._lin(p + "linear2.", p + "linear2.weight", p + "linear2.bias")
        for i in range(cfg.dec_layers):
            p = f"{vt}decoder.layers.{i}."
            self._mha(p + "self_attn.", split_qk=False)
            self._mha(p + "multihead_attn.", split_qk=True)
            self._lin(p + "linear1.", p + "linear1.weight", p + "linear1.bias")
            self._lin(p + "linear2.", p + "linear2.weight", p + "linear2.bias")
        for i in range(3):
            p = f"bbox_embed.layers.{i}."
            self._lin(p, p + "weight", p + "bias")
        if cfg.masks:                    # MHAttentionMap projections (reftr_segmentation.py:186-187)
            for n in ("q_linear.", "k_linear."):
                self._lin("bbox_attention." + n, "bbox_attention." + n + "weight", "bbox_attention." + n + "bias")
        # input_proj 1x1 conv [E, 2048, 1, 1] used as a Linear over pixels
        w = st.phys("input_proj.0.0.weight").view(E, 2048)
        gw = st.phys("input_proj.0.0.weight", grad=True).view(E, 2048)
        self.lins["input_proj.0.0."] = Lin(w, st.P["input_proj.0.0.bias"], gw, st.G["input_proj.0.0.bias"], st.device)
        for l in self.lins.values():             # every weight gradient of these goes through lin_bwd (which claims first writes)
            st.register_overwritable(l.gw)

    def _mha(self, p, split_qk):
        E = self.cfg.hidden
        wn, bn = p + "in_proj_weight", p + "in_proj_bias"
        if split_qk:
            self._lin(p + "q", wn, bn, rows=(0, E)); self._lin(p + "k", wn, bn, rows=(E, 2 * E))
        else:
            self._lin(p + "qk", wn, bn, rows=(0, 2 * E))
        self._lin(p + "v", wn, bn, rows=(2 * E, 3 * E))
        self._lin(p + "out_proj.", p + "out_proj.weight", p + "out_proj.bias")

    def refresh(self):
        if getattr(self, "_prep", None) is None:
            self._prep = H.WeightPrepBatch(self.store.device)
            for l in self.lins.values():
                self._prep.add(l.w32, l.N, 1, l.K, dst=l.W, dst_t=l.WT)
        self._prep.run()
        self._refresh_kv_cat()

    # ------------------------------------------------------------------ helpers
    def begin_step(self, training):
        self.training, self._site = training, 0

    def _drop(self, p):
        """(p, site id) of the next dropout site; p = 0 in eval mode.  The site id is mixed on the device with the
        model's step-seed word (hip.set_seed_dev), so a captured graph draws fresh masks at every replay."""
        self._site += 1
        if not self.training or p <= 0:
            return 0.0, 0
        return float(p), (self._site * 40503 + 977) & 0xFFFFFFFF

    def lin_fwd(self, key, x, **kw):
        l = self.lins[key]
        return H.linear(x, l.W, bias=l.b32, **kw)

    def lin_bwd(self, key, dy, x, need_dx=True, **kw):
        """weight + bias gradient of a Linear (accumulated) and, if asked, its input gradient."""
        l = self.lins[key]
        ow = self.store.claim(l.gw)                                # training loop: the first contribution overwrites
        if self.small_wg is not None and dy.shape[0] <= 16:       # decoder-side rows: queued, launched as one group
            self.small_wg.add(dy, x, l.gw, l.gb, overwrite=ow)
        elif self.big_wg is not None and dy.shape[0] > 16:
            self.big_wg.add(dy, x, l.gw, l.gb, overwrite=ow)       # flushed per section (flush_wgrads)
        else:
            self.wg.run(lambda: H.linear_wgrad(dy, x, l.gw, dbias=l.gb, overwrite=ow), dy, x)
        if need_dx:
            return H.linear(dy, l.WT, **kw)
        return None

    def flush_wgrads(self):
        """Launch every queued weight gradient (on the current stream; the queued dy / x tensors were produced on it)."""
        if self.small_wg is not None:
            self.small_wg.run()
        if self.big_wg is not None:
            self.big_wg.run()
        if self.ln_batch is not None:
            self.ln_batch.run()

    def flush_wgrads_side(self):
        """The same launches on the language stream, in front of the BERT-backward branch: the ResNet backward does not wait for them
        (-0.07 ms).  The queued tensors stay referenced by the SideStream until the join at the end of backward."""
        if not self.side.enabled:
            return self.flush_wgrads()
        keep = []
        for b in (self.small_wg, self.big_wg, self.ln_batch):
            for tup in getattr(b, "keep", []) if b is not None else []:
                keep.extend(t for t in tup if torch.is_tensor(t))
        if keep:
            self.side.run(self.flush_wgrads, *keep)

    def P(self, name):
        return self.store.P[name]

    def G(self, name):
        return self.store.G[name]

    def ln_fwd(self, x, pfx, eps=1e-5, **kw):
        return H.layernorm_fwd(x, self.P(pfx + "weight"), self.P(pfx + "bias"), eps, **kw)

    def ln_bwd(self, dy, x, pfx, mean, rstd, **kw):
        return H.layernorm_bwd(dy, x, self.P(pfx + "weight"), self.P(pfx + "bias"), mean, rstd,
                               self.G(pfx + "weight"), self.G(pfx + "bias"), pg_batch=self.ln_batch, **kw)

    # ------------------------------------------------------------------ BERT
    def bert_fwd(self, ids, mask_u8):
        """ids int64 [B, L], mask uint8 [B, L] (1 = token).  Returns (seq bf16 [B*L, Hd], pooled bf16 [B, Hd], ctx)."""
        bc = self.cfg.bert
        B, L = ids.shape
        M, Hd = B * L, bc.hidden
        pfx = "lang_backbone."
        e = pfx + "embeddings."
        kpm = (mask_u8 == 0).view(torch.uint8)
        pos_ids = H.roberta_pos_ids(ids, bc.pad_idx) if bc.pad_idx >= 0 else None
        emb = H.bert_embed_fwd(ids, self.P(e + "word_embeddings.weight"), self.P(e + "position_embeddings.weight"),
                               self.P(e + "token_type_embeddings.weight"), L, pos_ids=pos_ids)
        dp, ds = self._drop(bc.dropout)
        h32, h16, _, mean, rstd = self.ln_fwd(emb, e + "LayerNorm.", bc.eps, drop_p=dp, drop_seed=ds)
        ctx = {"ids": ids, "pos_ids": pos_ids, "kpm": kpm, "B": B, "L": L, "emb": emb, "emb_stats": (mean, rstd, dp, ds), "layers": []}
        dh = Hd // bc.heads
        scale = 1.0 / math.sqrt(dh)
        for i in range(bc.layers):
            lp = f"{pfx}encoder.layer.{i}."
            r = {"h16": h16}
            qkv, _ = self.lin_fwd(lp + "qkv", h16)
            r["qkv"] = qkv
            r["adrop"] = self._drop(bc.dropout)
            o, lse = H.attn_fwd(qkv[:, :Hd], qkv[:, Hd:2 * Hd], qkv[:, 2 * Hd:], kpm, B=B, H=bc.heads, Sq=L, Sk=L, dh=dh,
                                scale=scale, drop_p=r["adrop"][0], drop_seed=r["adrop"][1])
            r["o"], r["lse"] = o, lse
            r["d1"] = self._drop(bc.dropout)
            _, t = self.lin_fwd(lp + "attention.output.dense.", o, drop_p=r["d1"][0], drop_seed=r["d1"][1], res_f32=h32,
                                out_bf16=False, out_f32=True)
            h32, h16, _, m1, r1 = self.ln_fwd(t, lp + "attention.output.LayerNorm.", bc.eps)
            r["t"], r["st1"], r["h1_16"] = t, (m1, r1), h16
            g, _, u = self.lin_fwd(lp + "intermediate.dense.", h16, act=GELU, out_preact=True)
            r["g"], r["u"] = g, u
            r["d2"] = self._drop(bc.dropout)
            _, t2 = self.lin_fwd(lp + "output.dense.", g, drop_p=r["d2"][0], drop_seed=r["d2"][1], res_f32=h32,
                                 out_bf16=False, out_f32=True)
            h32, h16, _, m2, r2 = self.ln_fwd(t2, lp + "output.LayerNorm.", bc.eps)
            r["t2"], r["st2"] = t2, (m2, r2)
            ctx["layers"].append(r)
        cls = torch.empty(B, Hd, dtype=torch.bfloat16, device=h16.device)
        H.rows_add(B, Hd, a_bf16=h16, a_map=(1, L, 0), out_bf16=cls)
        pooled, _ = self.lin_fwd(pfx + "pooler.dense.", cls, act=TANH)
        ctx["cls"], ctx["pooled"] = cls, pooled
        return h16, pooled, ctx

    def bert_bwd(self, ctx, d_seq, d_pooled_bf16):
        """d_seq fp32 [B*L, Hd] = dL/d(sequence output) or None; d_pooled_bf16 [B, Hd] = dL/d(pooler PRE-tanh
        output) (callers fold tanh' with the GEMM's dtanh epilogue) or None."""
        for _ in self.bert_bwd_layers(ctx, d_seq, d_pooled_bf16):
            pass

    def bert_bwd_layers(self, ctx, d_seq, d_pooled_bf16, stops=()):
        """bert_bwd as a generator: yields i right after the backward of encoder layer i for every i in `stops`, with the
        weight gradients queued so far launched (the gradients of layers >= i and of the pooler are final then)."""
        bc = self.cfg.bert
        B, L, Hd = ctx["B"], ctx["L"], bc.hidden
        pfx = "lang_backbone."
        dev = ctx["emb"].device
        if d_seq is None:
            d_seq = torch.zeros(B * L, Hd, dtype=torch.float32, device=dev)
        if d_pooled_bf16 is not None:
            _, dcls = self.lin_bwd(pfx + "pooler.dense.", d_pooled_bf16, ctx["cls"], out_bf16=False, out_f32=True)
            H.rows_add(B, Hd, a_f32=dcls, out_f32=d_seq, accumulate=True, o_map=(1, L, 0))
        dh = Hd // bc.heads
        scale = 1.0 / math.sqrt(dh)
        dh32 = d_seq
        for i in reversed(range(bc.layers)):
            lp = f"{pfx}encoder.layer.{i}."
            r = ctx["layers"][i]
            dt2, dt2b = self.ln_bwd(dh32, r["t2"], lp + "output.LayerNorm.", *r["st2"], drop2_p=r["d2"][0], drop2_seed=r["d2"][1])
            du, _ = self.lin_bwd(lp + "output.dense.", dt2b, r["g"], preact=r["u"])
            _, dh1 = self.lin_bwd(lp + "intermediate.dense.", du, r["h1_16"], res_f32=dt2, out_bf16=False, out_f32=True)
            dt, dtb = self.ln_bwd(dh1, r["t"], lp + "attention.output.LayerNorm.", *r["st1"], drop2_p=r["d1"][0], drop2_seed=r["d1"][1])
            do, _ = self.lin_bwd(lp + "attention.output.dense.", dtb, r["o"])
            qkv = r["qkv"]
            dqkv = torch.empty_like(qkv)
            H.attn_bwd(qkv[:, :Hd], qkv[:, Hd:2 * Hd], qkv[:, 2 * Hd:], r["o"], do, r["lse"], ctx["kpm"], B=B, H=bc.heads,
                       Sq=L, Sk=L, dh=dh, scale=scale, drop_p=r["adrop"][0], drop_seed=r["adrop"][1],
                       dq=dqkv[:, :Hd], dk=dqkv[:, Hd:2 * Hd], dv=dqkv[:, 2 * Hd:])
            _, dh32 = self.lin_bwd(lp + "qkv", dqkv, r["h16"], res_f32=dt, out_bf16=False, out_f32=True)
            if i in stops:
                self.flush_wgrads()
                yield i
        e = pfx + "embeddings."
        mean, rstd, dp, ds = ctx["emb_stats"]
        de, _ = self.ln_bwd(dh32, ctx["emb"], e + "LayerNorm.", mean, rstd, drop_p=dp, drop_seed=ds, want_bf16=False)
        H.bert_embed_bwd(ctx["ids"], de, self.G(e + "word_embeddings.weight"), self.G(e + "position_embeddings.weight"),
                         self.G(e + "token_type_embeddings.weight"), L, pos_ids=ctx["pos_ids"])
        self.flush_wgrads()          # on this stream, and before a second BERT pass can queue the same weights

    # ------------------------------------------------------------------ mlp_mapping (reftr_transformer.py:14-23)
    def mlp_fwd(self, x16, pfx, **out_kw):
        """Linear-LN-ReLU-Dropout(.1)-Linear-LN-ReLU.  out_kw go to the final LayerNorm launch (output buffers,
        row map, pos).  Returns (outputs of that launch, ctx)."""
        _, t1 = self.lin_fwd(pfx + "0.", x16, out_bf16=False, out_f32=True)
        dp, ds = self._drop(0.1)
        _, a16, _, m1, r1 = self.ln_fwd(t1, pfx + "1.", act=RELU, drop_p=dp, drop_seed=ds, want_f32=False)
        _, t2 = self.lin_fwd(pfx + "4.", a16, out_bf16=False, out_f32=True)
        outs = self.ln_fwd(t2, pfx + "5.", act=RELU, **out_kw)
        ctx = {"x16": x16, "t1": t1, "st1": (m1, r1, dp, ds), "a16": a16, "t2": t2, "st2": (outs[3], outs[4])}
        return outs, ctx

    def mlp_bwd(self, ctx, dy, pfx, dy_rowmap=(0, 0, 0), dy2=None, need_dx=True, dx_f32=True, **dx_kw):
        _, dt2b = self.ln_bwd(dy, ctx["t2"], pfx + "5.", *ctx["st2"], act=RELU, rowmap=dy_rowmap, dy2=dy2, want_f32=False)
        _, da = self.lin_bwd(pfx + "4.", dt2b, ctx["a16"], out_bf16=False, out_f32=True)
        m1, r1, dp, ds = ctx["st1"]
        _, dt1b = self.ln_bwd(da, ctx["t1"], pfx + "1.", m1, r1, act=RELU, drop_p=dp, drop_seed=ds, want_f32=False)
        if not need_dx:
            self.lin_bwd(pfx + "0.", dt1b, ctx["x16"], need_dx=False)
            return None
        ob, of = self.lin_bwd(pfx + "0.", dt1b, ctx["x16"], out_bf16=not dx_f32, out_f32=dx_f32, **dx_kw)
        return of if dx_f32 else ob

    # ------------------------------------------------------------------ encoder layer (transformer.py:168-181)
    def enc_layer_fwd(self, p, x32, x16, xp16, pos, kpm, B, S, qkv=None, next_p=None):
        """One TransformerEncoderLayer (transformer.py:168-181).  `qkv` / `next_p`: unused since round 5 (they served the fused
        row-local launch rt_enc_tail_*, measured at the launched chain's speed in round 4 and removed: LAB_NOTES.md); the fifth return
        value is always None.  Returns (x2_32, x2_16, x2p16, saved, None)."""
        cfg = self.cfg
        E, Hh = cfg.hidden, cfg.nheads
        dh = E // Hh
        r = {"x16": x16, "xp16": xp16}
        if qkv is None:
            grp = H.GemmGroup()                      # two independent projections, one launch
            qk, _ = self.lin_fwd(p + "self_attn.qk", xp16, group=grp)
            v, _ = self.lin_fwd(p + "self_attn.v", x16, group=grp)
            grp.run()
        else:
            qk, v = qkv
        r["ad"] = self._drop(cfg.dropout)
        o, lse = H.attn_fwd(qk[:, :E], qk[:, E:], v, kpm, B=B, H=Hh, Sq=S, Sk=S, dh=dh, scale=dh ** -0.5,
                            drop_p=r["ad"][0], drop_seed=r["ad"][1])
        r.update(qk=qk, v=v, o=o, lse=lse)
        r["d1"] = self._drop(cfg.dropout)
        _, t = self.lin_fwd(p + "self_attn.out_proj.", o, drop_p=r["d1"][0], drop_seed=r["d1"][1], res_f32=x32,
                            out_bf16=False, out_f32=True)
        x1_32, x1_16, _, m1, r1 = self.ln_fwd(t, p + "norm1.")
        r.update(t=t, st1=(m1, r1), x1_16=x1_16)
        r["dh"] = self._drop(cfg.dropout)
        hdn, _ = self.lin_fwd(p + "linear1.", x1_16, act=RELU, drop_p=r["dh"][0], drop_seed=r["dh"][1])
        r["hdn"] = hdn
        r["d2"] = self._drop(cfg.dropout)
        _, t2 = self.lin_fwd(p + "linear2.", hdn, drop_p=r["d2"][0], drop_seed=r["d2"][1], res_f32=x1_32,
                             out_bf16=False, out_f32=True)
        x2_32, x2_16, x2p16, m2, r2 = self.ln_fwd(t2, p + "norm2.", pos=pos)
        r.update(t2=t2, st2=(m2, r2))
        return x2_32, x2_16, x2p16, r, None

    def enc_layer_bwd(self, p, r, dx2, dx2b, kpm, B, S, dpos_acc):
        """dx2 (+ optional dx2b) = gradient w.r.t. this layer's output.  Returns gradient w.r.t. its input x;
        the q/k-input gradient is also accumulated into dpos_acc (pos is re-added at every layer)."""
        cfg = self.cfg
        E, Hh = cfg.hidden, cfg.nheads
        dh = E // Hh
        M = B * S
        gs = 1.0 / (1.0 - r["dh"][0]) if r["dh"][0] > 0 else 1.0
        if getattr(self, "_dbg_enc", None) is not None:          # tests: the gradient that enters the layer
            self._dbg_enc.append((dx2 if dx2b is None else dx2 + dx2b).detach().clone())
        dt2, dt2b = self.ln_bwd(dx2, r["t2"], p + "norm2.", *r["st2"], dy2=dx2b, drop2_p=r["d2"][0], drop2_seed=r["d2"][1])
        dhdn, _ = self.lin_bwd(p + "linear2.", dt2b, r["hdn"], gate=r["hdn"], gate_scale=gs)
        _, dx1 = self.lin_bwd(p + "linear1.", dhdn, r["x1_16"], res_f32=dt2, out_bf16=False, out_f32=True)
        dt, dtb = self.ln_bwd(dx1, r["t"], p + "norm1.", *r["st1"], drop2_p=r["d1"][0], drop2_seed=r["d1"][1])
        do, _ = self.lin_bwd(p + "self_attn.out_proj.", dtb, r["o"])
        qk, v = r["qk"], r["v"]
        dqk = torch.empty_like(qk)
        _, _, dv = H.attn_bwd(qk[:, :E], qk[:, E:], v, r["o"], do, r["lse"], kpm, B=B, H=Hh, Sq=S, Sk=S, dh=dh,
                              scale=dh ** -0.5, drop_p=r["ad"][0], drop_seed=r["ad"][1], dq=dqk[:, :E], dk=dqk[:, E:])
        grp = H.GemmGroup()
        _, dxa = self.lin_bwd(p + "self_attn.v", dv, r["x16"], res_f32=dt, out_bf16=False, out_f32=True, group=grp)
        _, dxp = self.lin_bwd(p + "self_attn.qk", dqk, r["xp16"], out_bf16=False, out_f32=True, group=grp, acc2_f32=dpos_acc)
        grp.run()
        if getattr(self, "_dbg_enc", None) is not None:          # tests: the per-layer input gradient
            self._dbg_enc.append((dxa + dxp).detach().clone())
        return dxa, dxp          # sum of the two = gradient w.r.t. the layer input

    # ------------------------------------------------------------------ decoder layer (transformer.py:231-252)
    def dec_kv_all(self, prefixes, mem16, memp16):
        """The cross-attention K / V projections of EVERY decoder layer (they only depend on the encoder memory,
        transformer.py:231-252): 2 x layers products in one launch, off the decoder's query chain."""
        grp = H.GemmGroup()
        out = []
        for p in prefixes:
            k2, _ = self.lin_fwd(p + "multihead_attn.k", memp16, group=grp)
            v2, _ = self.lin_fwd(p + "multihead_attn.v", mem16, group=grp)
            out.append((k2, v2))
        grp.run()
        return out

    def dec_layer_fwd(self, p, t32, t16, tq16, qpos, mem16, memp16, qmask, kpm, B, T, S, kv=None, t3_out=None, fold_sa=False):
        cfg = self.cfg
        E, Hh = cfg.hidden, cfg.nheads
        dh = E // Hh
        sc = dh ** -0.5
        r = {"t16": t16, "tq16": tq16}
        # One query per image and no padded phrase: the self-attention softmax runs over a single key, so it is exactly 1
        # whatever q and k are, and dq = dk = 0 exactly -- the q/k projection, its backward and its (zero) weight gradient
        # are skipped.  What is left of the attention is its per-head probability dropout on the value path: with fold_sa
        # (single-phrase inputs: the one key is never padding) that is the V projection's own epilogue (one keep/drop decision
        # per head_dim features, same seed site and hash index b * H + h as the attention kernel), else the attention kernel.
        trivial = T == 1 and self.trivial_sa
        fold = trivial and fold_sa and self.fold_sa and (dh & (dh - 1)) == 0
        r["trivial"], r["fold"] = trivial, fold
        lse = None
        if fold:
            r["ad"] = self._drop(cfg.dropout)
            qk = v = None
            o, _ = self.lin_fwd(p + "self_attn.v", t16, drop_p=r["ad"][0], drop_seed=r["ad"][1], drop_shift=dh.bit_length() - 1)
        else:
            v, _ = self.lin_fwd(p + "self_attn.v", t16)
            r["ad"] = self._drop(cfg.dropout)
            if trivial:
                qk = None
                o, lse = H.attn_fwd(v, v, v, qmask, B=B, H=Hh, Sq=1, Sk=1, dh=dh, scale=sc, drop_p=r["ad"][0], drop_seed=r["ad"][1])
            else:
                qk, _ = self.lin_fwd(p + "self_attn.qk", tq16)
                o, lse = H.attn_fwd(qk[:, :E], qk[:, E:], v, qmask, B=B, H=Hh, Sq=T, Sk=T, dh=dh, scale=sc,
                                    drop_p=r["ad"][0], drop_seed=r["ad"][1])
        r.update(qk=qk, v=v, o=o, lse=lse)
        r["d1"] = self._drop(cfg.dropout)
        _, u = self.lin_fwd(p + "self_attn.out_proj.", o, drop_p=r["d1"][0], drop_seed=r["d1"][1], res_f32=t32,
                            out_bf16=False, out_f32=True)
        t1_32, t1_16, t1q16, m1, r1 = self.ln_fwd(u, p + "norm1.", pos=qpos)
        r.update(u=u, st1=(m1, r1), t1q16=t1q16)
        q2, _ = self.lin_fwd(p + "multihead_attn.q", t1q16)
        if kv is not None:
            k2, v2 = kv
        else:
            k2, _ = self.lin_fwd(p + "multihead_attn.k", memp16)
            v2, _ = self.lin_fwd(p + "multihead_attn.v", mem16)
        r["ad2"] = self._drop(cfg.dropout)
        o2, lse2 = H.attn_fwd(q2, k2, v2, kpm, B=B, H=Hh, Sq=T, Sk=S, dh=dh, scale=sc, drop_p=r["ad2"][0], drop_seed=r["ad2"][1])
        r.update(q2=q2, k2=k2, v2=v2, o2=o2, lse2=lse2)
        r["d2"] = self._drop(cfg.dropout)
        _, u2 = self.lin_fwd(p + "multihead_attn.out_proj.", o2, drop_p=r["d2"][0], drop_seed=r["d2"][1], res_f32=t1_32,
                             out_bf16=False, out_f32=True)
        t2_32, t2_16, _, m2, r2 = self.ln_fwd(u2, p + "norm2.")
        r.update(u2=u2, st2=(m2, r2), t2_16=t2_16)
        r["dh"] = self._drop(cfg.dropout)
        hdn, _ = self.lin_fwd(p + "linear1.", t2_16, act=RELU, drop_p=r["dh"][0], drop_seed=r["dh"][1])
        r["hdn"] = hdn
        r["d3"] = self._drop(cfg.dropout)
        _, u3 = self.lin_fwd(p + "linear2.", hdn, drop_p=r["d3"][0], drop_seed=r["d3"][1], res_f32=t2_32,
                             out_bf16=False, out_f32=True)
        t3_32, t3_16, t3q16, m3, r3 = self.ln_fwd(u3, p + "norm3.", pos=qpos, y_f32=t3_out)
        r.update(u3=u3, st3=(m3, r3))
        return t3_32, t3_16, t3q16, r

    def _dec_handoff_buf(self, dev):
        if self._dec_handoff is None or self._dec_handoff.device != dev:
            self._dec_handoff = H.decoder_handoff(dev)          # this model's: its launches are ordered on its stream
        return self._dec_handoff

    def dec_stack_coop_ok(self, N, T, S, n_layers, fold_sa):
        """Shapes rt_decoder_fwd covers: one query per image on the folded self-attention path, the reference's widths."""
        cfg = self.cfg
        if self._dec_cus is None:
            # the launches need their ~80 spin-waiting workgroups resident at once: the library answers from the occupancy query
            # and the device's compute-unit count (rt_decoder_supported), per device, once
            dev = self.store.device
            ok = False
            if str(dev).startswith("cuda"):
                with torch.cuda.device(dev):
                    ok = H.decoder_supported(cfg.ffn)
            self._dec_cus = 1 << 20 if ok else 0
        return (self.dec_coop and self._dec_cus >= 160 and T == 1 and fold_sa and self.trivial_sa and self.fold_sa and cfg.hidden == 256 and cfg.nheads == 8
                and cfg.ffn == 2048 and N <= 16 and S <= 768 and 1 <= n_layers <= H.DEC_MAX_LAYERS)

    def dec_stack_fwd_coop(self, prefixes, t32, t16, tq16, qpos, kvs, kpm, B, S, t3_all):
        """All decoder layers in ONE launch (rt_decoder_fwd).  Writes exactly the tensors the per-layer chain (dec_layer_fwd,
        folded path) saves -- same values bit for bit -- and returns the same per-layer dicts, so dec_layer_bwd is unchanged."""
        cfg = self.cfg
        E, F, N = cfg.hidden, cfg.ffn, t32.shape[0]
        dev = t32.device
        bf, f32 = torch.bfloat16, torch.float32
        layers, saved = [], []
        t16_in = t16
        for i, p in enumerate(prefixes):
            r = {"t16": t16_in, "tq16": tq16 if i == 0 else None, "trivial": True, "fold": True, "qk": None, "v": None, "lse": None,
                 "coop": True}
            for k in ("ad", "d1", "ad2", "d2", "dh", "d3"):       # the chain's site order
                r[k] = self._drop(cfg.dropout)
            sm = torch.empty(6, N, dtype=f32, device=dev)         # mean / rstd of norm1..3
            act = torch.empty(5, N, E, dtype=bf, device=dev)      # o, t1q16, q2, o2, t2_16
            pre = torch.empty(3, N, E, dtype=f32, device=dev)     # u, u2, u3
            t3_16 = torch.empty(N, E, dtype=bf, device=dev)
            hdn = torch.empty(N, F, dtype=bf, device=dev)
            lse2 = torch.empty(B, cfg.nheads, 1, dtype=f32, device=dev)
            k2, v2 = kvs[i]
            r.update(o=act[0], t1q16=act[1], q2=act[2], o2=act[3], t2_16=act[4], u=pre[0], u2=pre[1], u3=pre[2], hdn=hdn,
                     st1=(sm[0], sm[1]), st2=(sm[2], sm[3]), st3=(sm[4], sm[5]), k2=k2, v2=v2, lse2=lse2)
            L = self.lins
            lay = dict(Wv=L[p + "self_attn.v"].W, Wo=L[p + "self_attn.out_proj."].W, Wq=L[p + "multihead_attn.q"].W,
                       Wo2=L[p + "multihead_attn.out_proj."].W, W1=L[p + "linear1."].W, W2=L[p + "linear2."].W,
                       bv=L[p + "self_attn.v"].b32, bo=L[p + "self_attn.out_proj."].b32, bq=L[p + "multihead_attn.q"].b32,
                       bo2=L[p + "multihead_attn.out_proj."].b32, b1=L[p + "linear1."].b32, b2=L[p + "linear2."].b32,
                       g1=self.P(p + "norm1.weight"), be1=self.P(p + "norm1.bias"), g2=self.P(p + "norm2.weight"),
                       be2=self.P(p + "norm2.bias"), g3=self.P(p + "norm3.weight"), be3=self.P(p + "norm3.bias"),
                       k2=k2, v2=v2, o=act[0], t1q16=act[1], q2=act[2], o2=act[3], t2_16=act[4], hdn=hdn, t3_16=t3_16,
                       u=pre[0], u2=pre[1], u3=pre[2], mean1=sm[0], rstd1=sm[1], mean2=sm[2], rstd2=sm[3], mean3=sm[4], rstd3=sm[5],
                       lse2=lse2, t3_f32=t3_all[i * N:(i + 1) * N],
                       seed_ad=r["ad"][1], seed_d1=r["d1"][1], seed_ad2=r["ad2"][1], seed_d2=r["d2"][1], seed_dh=r["dh"][1],
                       seed_d3=r["d3"][1])
            layers.append(lay); saved.append(r)
            t16_in = t3_16
        dh = E // cfg.nheads
        drop_p = saved[0]["ad"][0]
        self.dec_counters = H.decoder_fwd(layers, t32, t16, qpos, kpm, H=cfg.nheads, S=S, F=F, drop_p=drop_p, scale=dh ** -0.5,
                                          handoff=self._dec_handoff_buf(dev))
        return saved

    def _wgrad_only(self, key, dy, x):
        """The weight / bias gradient half of lin_bwd (the backward-data half ran elsewhere)."""
        l = self.lins[key]
        ow = self.store.claim(l.gw)
        if self.small_wg is not None and dy.shape[0] <= 16:
            self.small_wg.add(dy, x, l.gw, l.gb, overwrite=ow)
        elif self.big_wg is not None and dy.shape[0] > 16:
            self.big_wg.add(dy, x, l.gw, l.gb, overwrite=ow)
        else:
            self.wg.run(lambda: H.linear_wgrad(dy, x, l.gw, dbias=l.gb, overwrite=ow), dy, x)

    def _dec_kv_wt_cat(self, prefixes):
        """[Wv_0^T | Wv_1^T | ...] and the same for K: the backward-data operands of every layer's cross-attention K / V projection
        side by side along the contraction axis, so that d memory = sum over layers of dV_l Wv_l is ONE product over K = layers x 256
        (refreshed with the other operands, see refresh)."""
        if self._kv_cat is None or self._kv_cat[0] != tuple(prefixes):
            E = self.cfg.hidden
            dev = self.store.device
            bufs = {n: torch.empty(E, len(prefixes) * E, dtype=torch.bfloat16, device=dev) for n in ("k", "v")}
            self._kv_cat = (tuple(prefixes), bufs)
            self._refresh_kv_cat()
        return self._kv_cat[1]

    def _refresh_kv_cat(self):
        if self._kv_cat is not None:
            prefixes, bufs = self._kv_cat
            for n in ("k", "v"):
                torch.cat([self.lins[p + "multihead_attn." + n].WT for p in prefixes], dim=1, out=bufs[n])

    def dec_stack_bwd_coop(self, prefixes, saved, dnorm_all, mem16, memp16, kpm, B, S, dmem_acc, dmemp_acc, dqpos_acc):
        """Backward of every decoder layer in ONE launch (rt_decoder_bwd) + what stays outside it, queued / launched exactly as
        dec_layer_bwd would: the grouped M <= 16 weight gradients and LayerNorm parameter gradients (their dy operands and
        partial sums come from the launch), the memory-gradient products and weight gradients of the cross-attention K / V
        projections.  Returns the gradient w.r.t. the stack's input."""
        cfg = self.cfg
        E, F, N = cfg.hidden, cfg.ffn, B
        dev = dnorm_all.device
        bf, f32 = torch.bfloat16, torch.float32
        nb = (N + 3) // 4
        layers, outs = [], []
        NLd = len(prefixes)
        pack = self.dec_kv_pack and NLd > 1
        dkv_all = torch.empty(2, B * S, NLd * E, dtype=bf, device=dev) if pack else None       # [dK_0 | dK_1 | ...], [dV_0 | ...]
        for i, p in enumerate(prefixes):
            r = saved[i]
            d16 = torch.empty(5, N, E, dtype=bf, device=dev)          # du3b, du2b, dq2, dub, dv
            dhdn = torch.empty(N, F, dtype=bf, device=dev)
            dkv = torch.empty(2, B * S, E, dtype=bf, device=dev)
            parts = torch.empty(3, nb, 2, E, dtype=f32, device=dev)
            L = self.lins
            lay = dict(WT2=L[p + "linear2."].WT, WT1=L[p + "linear1."].WT, WTo2=L[p + "multihead_attn.out_proj."].WT,
                       WTq=L[p + "multihead_attn.q"].WT, WTo=L[p + "self_attn.out_proj."].WT, WTv=L[p + "self_attn.v"].WT,
                       g1=self.P(p + "norm1.weight"), g2=self.P(p + "norm2.weight"), g3=self.P(p + "norm3.weight"),
                       u=r["u"], u2=r["u2"], u3=r["u3"], mean1=r["st1"][0], rstd1=r["st1"][1], mean2=r["st2"][0], rstd2=r["st2"][1],
                       mean3=r["st3"][0], rstd3=r["st3"][1], hdn=r["hdn"], q2=r["q2"], k2=r["k2"], v2=r["v2"], o2=r["o2"],
                       lse2=r["lse2"], dnorm=dnorm_all[i * N:(i + 1) * N], du3b=d16[0], dhdn=dhdn, du2b=d16[1], dq2=d16[2],
                       dub=d16[3], dv=d16[4], dk2=dkv[0], dv2=dkv[1], part1=parts[0], part2=parts[1], part3=parts[2],
                       seed_ad=r["ad"][1], seed_d1=r["d1"][1], seed_ad2=r["ad2"][1], seed_d2=r["d2"][1], seed_d3=r["d3"][1])
            if pack:
                lay.update(dk2p=dkv_all[0][:, i * E:(i + 1) * E], dv2p=dkv_all[1][:, i * E:(i + 1) * E])
            layers.append(lay); outs.append((d16, dhdn, dkv, parts))
        drop_p = saved[0]["d3"][0]
        p_dh = saved[0]["dh"][0]
        dta = torch.empty(N, E, dtype=f32, device=dev)
        dh = E // cfg.nheads
        self.dec_counters = H.decoder_bwd(layers, dta, dqpos_acc, kpm, H=cfg.nheads, S=S, F=F, drop_p=drop_p, scale=dh ** -0.5,
                                          gate_scale=1.0 / (1.0 - p_dh) if p_dh > 0 else 1.0, ldkvp=NLd * E if pack else 0,
                                          handoff=self._dec_handoff_buf(dev))
        for i in reversed(range(len(prefixes))):
            p, r = prefixes[i], saved[i]
            d16, dhdn, dkv, parts = outs[i]
            self._wgrad_only(p + "linear2.", d16[0], r["hdn"])
            self._wgrad_only(p + "linear1.", dhdn, r["t2_16"])
            self._wgrad_only(p + "multihead_attn.out_proj.", d16[1], r["o2"])
            if pack:                                 # weight gradients queued; the memory gradient is one product below
                self._wgrad_only(p + "multihead_attn.v", dkv[1], mem16)
                self._wgrad_only(p + "multihead_attn.k", dkv[0], memp16)
            else:
                grp = H.GemmGroup()                  # the M = B*S products: regular launches, as in dec_layer_bwd
                self.lin_bwd(p + "multihead_attn.v", dkv[1], mem16, res_f32=dmem_acc, out_bf16=False, out_f32=dmem_acc, group=grp)
                self.lin_bwd(p + "multihead_attn.k", dkv[0], memp16, res_f32=dmemp_acc, out_bf16=False, out_f32=dmemp_acc, group=grp)
                grp.run()
            self._wgrad_only(p + "multihead_attn.q", d16[2], r["t1q16"])
            self._wgrad_only(p + "self_attn.out_proj.", d16[3], r["o"])
            self._wgrad_only(p + "self_attn.v", d16[4], r["t16"])
            for j, nm in ((2, "norm3."), (1, "norm2."), (0, "norm1.")):
                self.ln_batch.jobs.append(H.LnPgJob(parts[j].data_ptr(), self.G(p + nm + "weight").data_ptr(),
                                                    self.G(p + nm + "bias").data_ptr(), nb, E))
                self.ln_batch.keep.append(parts[j])
        if pack:
            # d memory += [dV_0 | dV_1 | ...] [Wv_0^T | Wv_1^T | ...]^T (and K with memory + pos): one launch, K = layers x 256, instead
            # of one read-modify-write launch per layer behind the cooperative one
            cat = self._dec_kv_wt_cat(prefixes)
            grp = H.GemmGroup()
            H.linear(dkv_all[1], cat["v"], res_f32=dmem_acc, out_bf16=False, out_f32=dmem_acc, group=grp)
            H.linear(dkv_all[0], cat["k"], res_f32=dmemp_acc, out_bf16=False, out_f32=dmemp_acc, group=grp)
            grp.run()
        return dta

    def dec_layer_bwd(self, p, r, g_a, g_b, mem16, memp16, qmask, kpm, B, T, S, dmem_acc, dmemp_acc, dqpos_acc):
        """g_a (+ g_b) = gradient w.r.t. this layer's output t3.  Returns (dt_a, dt_q): their sum is the
        gradient w.r.t. the layer input t; dt_q (gradient w.r.t. t + query_pos) is also added to dqpos_acc."""
        cfg = self.cfg
        E, Hh = cfg.hidden, cfg.nheads
        dh = E // Hh
        sc = dh ** -0.5
        N = B * T
        du3, du3b = self.ln_bwd(g_a, r["u3"], p + "norm3.", *r["st3"], dy2=g_b, drop2_p=r["d3"][0], drop2_seed=r["d3"][1])
        gs = 1.0 / (1.0 - r["dh"][0]) if r["dh"][0] > 0 else 1.0
        dhdn, _ = self.lin_bwd(p + "linear2.", du3b, r["hdn"], gate=r["hdn"], gate_scale=gs)
        _, dt2 = self.lin_bwd(p + "linear1.", dhdn, r["t2_16"], res_f32=du3, out_bf16=False, out_f32=True)
        du2, du2b = self.ln_bwd(dt2, r["u2"], p + "norm2.", *r["st2"], drop2_p=r["d2"][0], drop2_seed=r["d2"][1])
        do2, _ = self.lin_bwd(p + "multihead_attn.out_proj.", du2b, r["o2"])
        dq2, dk2, dv2 = H.attn_bwd(r["q2"], r["k2"], r["v2"], r["o2"], do2, r["lse2"], kpm, B=B, H=Hh, Sq=T, Sk=S, dh=dh,
                                   scale=sc, drop_p=r["ad2"][0], drop_seed=r["ad2"][1])
        grp = H.GemmGroup()                      # two accumulators, two independent products, one launch
        self.lin_bwd(p + "multihead_attn.v", dv2, mem16, res_f32=dmem_acc, out_bf16=False, out_f32=dmem_acc, group=grp)
        self.lin_bwd(p + "multihead_attn.k", dk2, memp16, res_f32=dmemp_acc, out_bf16=False, out_f32=dmemp_acc, group=grp)
        # (on the language stream, which idles here, this M = B*S launch delays the chain's 1-16-workgroup kernels by more than leaving
        # the chain saves: +0.2 ms, profiles/r03_side_stream_probes.txt)
        grp.run()
        _, dt1q = self.lin_bwd(p + "multihead_attn.q", dq2, r["t1q16"], out_bf16=False, out_f32=True, acc2_f32=dqpos_acc)
        du, dub = self.ln_bwd(du2, r["u"], p + "norm1.", *r["st1"], dy2=dt1q, drop2_p=r["d1"][0], drop2_seed=r["d1"][1])
        if r["fold"]:           # the head-dropout mask of the forward, applied by the backward-data product's epilogue
            dv, _ = self.lin_bwd(p + "self_attn.out_proj.", dub, r["o"], drop_p=r["ad"][0], drop_seed=r["ad"][1],
                                 drop_shift=dh.bit_length() - 1)
            _, dta = self.lin_bwd(p + "self_attn.v", dv, r["t16"], res_f32=du, out_bf16=False, out_f32=True)
            return dta, None
        do, _ = self.lin_bwd(p + "self_attn.out_proj.", dub, r["o"])
        qk, v = r["qk"], r["v"]
        if r["trivial"]:
            _, _, dv = H.attn_bwd(v, v, v, r["o"], do, r["lse"], qmask, B=B, H=Hh, Sq=1, Sk=1, dh=dh,
                                  scale=sc, drop_p=r["ad"][0], drop_seed=r["ad"][1])
            _, dta = self.lin_bwd(p + "self_attn.v", dv, r["t16"], res_f32=du, out_bf16=False, out_f32=True)
            return dta, None
        dqk = torch.empty_like(qk)
        _, _, dv = H.attn_bwd(qk[:, :E], qk[:, E:], v, r["o"], do, r["lse"], qmask, B=B, H=Hh, Sq=T, Sk=T, dh=dh,
                              scale=sc, drop_p=r["ad"][0], drop_seed=r["ad"][1], dq=dqk[:, :E], dk=dqk[:, E:])
        _, dta = self.lin_bwd(p + "self_attn.v", dv, r["t16"], res_f32=du, out_bf16=False, out_f32=True)
        _, dtq = self.lin_bwd(p + "self_attn.qk", dqk, r["tq16"], out_bf16=False, out_f32=True, acc2_f32=dqpos_acc)
        return dta, dtq
