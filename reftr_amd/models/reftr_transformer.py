"""RefTR (transformer_single_phrase / multi-phrase REC model) on the MI355X kernel library.

Host-side mirror of the reference's models/reftr_transformer.py: same constructor role, same
`forward(samples) -> {'pred_boxes', 'phrase_mask', 'aux_outputs'}` contract (:159-304), same state_dict
keys (built from models/layout.py), same `build_reftr(args)` factory (:307-347).  Underneath, forward and
backward are explicit sequences of libreftr_hip.so launches (backbone.py / net.py); torch.autograd only
sees ONE node for the whole model (`_RefTRFunction`), whose backward runs the hand-written backward pass
and leaves every parameter gradient in the flat gradient buffer.
"""
import math
import os

import torch
from torch import nn

from .. import hip as H
from ..util.misc import NestedTensor, nested_tensor_from_tensor_list
from . import layout as L
from .backbone import ResNetBody
from .net import Net, RELU
from .segmentation import SegHead
from .store import ParamStore, build_module_tree, rebind


class _RefTRFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, samples):
        ctx.model = model
        return model._forward_impl(samples)

    @staticmethod
    def backward(ctx, dlogits):
        ctx.model._backward_impl(dlogits.contiguous())
        return None, None, None


class _RefTRSegFunction(torch.autograd.Function):
    """RefTRSeg: two differentiable outputs (box logits, mask logits), one hand-written backward."""

    @staticmethod
    def forward(ctx, anchor, model, samples):
        ctx.model = model
        logits = model._forward_impl(samples)
        if model.cfg.cem:
            return logits, model._saved["pred_masks"], model._saved["seg"]["cem_loss"]
        return logits, model._saved["pred_masks"]

    @staticmethod
    def backward(ctx, dlogits, dmasks, dcem=None):
        ctx.model._backward_impl(dlogits.contiguous(), dmasks.contiguous(), None if dcem is None else dcem.contiguous())
        return None, None, None


class RefTR(nn.Module):
    def __init__(self, cfg: L.ModelConfig, device="cuda", aux_loss=True):
        super().__init__()
        assert cfg.n_q >= 1
        assert cfg.n_q == 1 or not cfg.masks, "RefTRSeg predicts one mask per image (n_ph = n_q = 1, reftr_segmentation.py:101-103)"
        self.cfg = cfg
        self.aux_loss = aux_loss and not cfg.masks       # RefTRSeg is built with aux_loss=False (reftr_segmentation.py:52)
        self.num_queries_per_phrase = cfg.n_q
        self.hidden_dim = cfg.hidden
        self.store = ParamStore(cfg, device)
        build_module_tree(self, self.store)
        self.body = ResNetBody(self.store, cfg)
        self.net = Net(self.store, cfg)
        self.seg = SegHead(self.store, cfg, self.net) if cfg.masks else None
        self._anchor = torch.zeros((), device=device, requires_grad=True)
        self.seed_dev = torch.zeros(1, dtype=torch.int32, device=device)   # dropout step-seed (advanced on device)
        self._operands_dirty = True       # bf16 operands must be rebuilt (after init / load / optimizer step)
        self._full_refresh = True
        self._step = 0
        self._saved = None
        # Data-parallel exchange points of backward, in the order they are reached (BOUNDARIES): at each one a slice of the
        # flat gradient buffer is final.  `_phase_hooks[name]` run there (the eager loop's asynchronous all-reduces);
        # `_stops` makes backward return there instead (CapturedTrainStep captures one graph per segment and issues the
        # collectives between replays; `continue_backward()` resumes).
        self._phase_hooks = {}
        # REFTR_DDP_SCHEDULE: "interleave" (default) -- BERT's backward stays on the language stream, one third beside each
        # ResNet stage, the two slices of a pair exchanged together; "serial" -- BERT's backward on the main stream in front of
        # the ResNet's (every BERT byte is exchanged under the ResNet backward, at the price of 1.6 ms of serialised compute)
        self.dp_schedule = os.environ.get("REFTR_DDP_SCHEDULE", "interleave")
        assert self.dp_schedule in ("interleave", "serial")
        self._stops = frozenset()
        self._bwd_gen = None
        self._slice_bounds = None         # set by the data-parallel wrapper: {boundary: slice(s) of flat_g that are final there}
        self._post_backward_hooks = []
        self._lin_refresh_pending = False
        # deferred optimizer (engine_vg.CapturedTrainStep): (main-stream fn, language-stream fn) that apply the previous
        # iteration's AdamW update at the head of this forward -- the BERT slice on the language stream, concurrently with
        # the ResNet forward -- and the engine's callback that applies a still-pending update before anyone else reads
        self._pre_update = None
        # single-process captured training step: the BERT slice's share of the gradient norm is taken on the language stream as
        # soon as that slice is final (see _backward_gen); (begin, end, device scalar) for the optimizer, None when not taken.
        # Anything that edits the gradient buffer between backward and the clip (gradient surgery, an exchange) must leave it off.
        # (Rounds 3-4 measured four other placements of the deferred AdamW passes -- beside the frozen prefix (REFTR_PRE_SIDE), behind it
        # (REFTR_OPT_LATE), serialised in front of it (REFTR_OPT_SERIAL), in five pieces on a third stream (REFTR_OPT_PIPE) -- all
        # neutral or slower (profiles/r03_side_stream_probes.txt, r04s_*, r04z_*); their code was removed in round 5, LAB_NOTES.md.)
        self._lang_tail = True            # map_sentence / map_phrase (forward and backward) on the language stream (r04ax, r04az)
        self._bb_ready = None
        self._norm_side = False          # switched on by engine_vg.CapturedTrainStep around its own backward + clip-norm unit only
        self._norm_split = None
        self._sq_bert = torch.zeros(1, dtype=torch.float32, device=device)
        self._flush_pending = None
        self._pending = None
        self.reset_parameters()

    # ------------------------------------------------------------------ init / state
    @torch.no_grad()
    def reset_parameters(self, seed=0):
        """Random init with the reference's distributions (models/reftr.py:45-49 xavier on every VLTransformer
        matrix, level_embed ~ N(0,1); reftr_transformer.py:131-135 zero last bbox layer, xavier input_proj;
        BERT-style N(0, 0.02); kaiming for convs; identity FrozenBN; RES head: xavier projections, kaiming_uniform(a=1)
        convolutions with zero bias, unit GroupNorm, reftr_segmentation.py:190-193,238-241)."""
        g = torch.Generator(device="cpu").manual_seed(seed)
        for name, shape, kind in self.store.table:
            t = self.store.P[name]
            leaf = name.rsplit(".", 1)[-1]
            if kind == "buffer":
                t.fill_(1.0 if leaf in ("weight", "running_var") else 0.0)
                continue
            is_norm = ("LayerNorm" in name or ".norm" in name or name.startswith("input_proj.0.1")
                       or (name.startswith("mask_head.gn"))       # GroupNorm of MaskHeadSmallConv: weight 1, bias 0 (torch default)
                       or (len(shape) == 1 and name.split(".")[-2] in ("1", "5") and leaf in ("weight", "bias")
                           and any(s in name for s in ("map_sentence", "map_phrase", "fuse_encoder_query", "context_out"))))
            if is_norm:
                t.fill_(1.0 if leaf == "weight" else 0.0)
            elif len(shape) == 1:
                t.zero_()
            else:
                fan_out, fan_in = shape[0], 1
                for s in shape[1:]:
                    fan_in *= s
                rf = 1
                for s in shape[2:]:
                    rf *= s
                v = torch.empty(shape, dtype=torch.float32)
                if name.startswith("lang_backbone."):
                    v.normal_(0.0, 0.02, generator=g)
                elif name.startswith("img_backbone.1."):
                    v.uniform_(0.0, 1.0, generator=g)                     # PositionEmbeddingLearned.reset_parameters, position_encoding.py:70-72
                elif name.startswith("img_backbone."):
                    v.normal_(0.0, math.sqrt(2.0 / (fan_out * rf)), generator=g)
                elif name == "vl_transformer.level_embed":
                    v.normal_(0.0, 1.0, generator=g)
                elif name.startswith("vl_transformer.") or name.startswith("input_proj.0.0") or name.startswith("bbox_attention."):
                    a = math.sqrt(6.0 / (fan_in + fan_out * rf))          # xavier_uniform (also reftr_segmentation.py:190-193)
                    v.uniform_(-a, a, generator=g)
                elif name.startswith("mask_head."):
                    a = math.sqrt(3.0 / fan_in)                           # kaiming_uniform_(a=1), reftr_segmentation.py:238-241
                    v.uniform_(-a, a, generator=g)
                else:
                    a = 1.0 / math.sqrt(fan_in)
                    v.uniform_(-a, a, generator=g)
                t.copy_(v)
        self.store.P["bbox_embed.layers.2.weight"].zero_()
        self.store.P["bbox_embed.layers.2.bias"].zero_()
        self.mark_dirty(full=True)

    def mark_dirty(self, full=False):
        self._operands_dirty = True
        self._full_refresh = self._full_refresh or full

    def operand_jobs(self):
        """Every trainable weight matrix as (element offset in the flat parameter buffer, N, T, C, FrozenBN scale | None, bf16
        operand [N][T][C], bf16 transposed operand [C][T][N]): what the optimizer's matrix-aware pass (rt_adamw_mat) needs to write
        the GEMM operands itself.  None until the operands exist (they are allocated by the first full refresh)."""
        if self._full_refresh or not self.body.W:
            return None
        st, fp = self.store, self.store.flat_p

        def off(t):
            o = (t.data_ptr() - fp.data_ptr()) // 4
            assert 0 <= o and o + t.numel() <= fp.numel() and t.is_contiguous()
            return o
        jobs = [(off(l.w32), l.N, 1, l.K, None, l.W, l.WT) for l in self.net.lins.values()]
        for c in self.body.all_convs:
            if c.trainable:
                jobs.append((off(st.phys(c.name)), c.cout, c.k * c.k, c.cin, self.body.bn[c.bn][0], self.body.W[c.name],
                             self.body.W[c.name + ".t"]))
        if self.seg is not None:
            jobs += [(off(c.w32), c.cop, c.k * c.k, c.cip, None, c.W, c.WT) for c in self.seg.convs.values()]
        # large tables that feed no GEMM (BERT's word / position embeddings: 24 M parameters) take the same tiled pass without
        # operand copies: it streams at 5.5 TB/s where the chunk pass over them ran at 3.9
        spans = sorted((j[0], j[0] + j[1] * j[2] * j[3]) for j in jobs)
        import bisect
        for name, shape, kind in st.table:
            if kind != "param" or len(shape) != 2 or name in st.physd:
                continue
            n = shape[0] * shape[1]
            o = st.offset[name][1]
            i = bisect.bisect_right(spans, (o, 1 << 62))          # first job that starts behind o; the one in front may still cover it
            covered = (i < len(spans) and spans[i][0] < o + n) or (i > 0 and spans[i - 1][1] > o)
            if n >= 65536 and shape[1] % 4 == 0 and not covered:
                jobs.append((o, shape[0], 1, shape[1], None, None, None))
        return jobs

    def refresh_now(self):
        """Rebuilds whatever is dirty on the current stream, now (callers that changed the masters outside an optimizer step and
        then replay a captured graph: the graph itself no longer contains an operand refresh)."""
        self.refresh_operands()
        if self._lin_refresh_pending:
            self.net.refresh()
            self._lin_refresh_pending = False

    def coop_failure_word(self):
        """The device word the cooperative decoder launches raise when a stage hand-off timed out (1-element int32 view), or None
        when this model cannot launch them."""
        net = self.net
        if not (net.dec_coop or net.dec_coop_bwd) or net._dec_handoff is None:
            return None
        return net._dec_handoff[1:2]

    def operands_emitted(self):
        """The optimizer's pass wrote every bf16 operand itself: nothing to rebuild except the K-concatenated copies."""
        self.net._refresh_kv_cat()

    def refresh_operands(self):
        """fp32 masters -> bf16 GEMM operands (weights [N,K] and [K,N], BN-folded conv weights)."""
        if not self._operands_dirty:
            return
        self.body.refresh(self._full_refresh)
        # the Linear operands (BERT + transformer, 85 % of the bytes) are refreshed at the head of the BERT side stream
        # inside _forward_impl, concurrently with the stem / layer1 of the ResNet
        self._lin_refresh_pending = True
        if self.seg is not None:
            self.seg.refresh()
        self._operands_dirty, self._full_refresh = False, False

    def state_dict(self, *args, **kwargs):
        if self._flush_pending is not None:
            self._flush_pending()
        sd = super().state_dict(*args, **kwargs)
        for k in list(sd.keys()):     # views of the flat buffer -> standalone, contiguous tensors
            sd[k] = sd[k].detach().clone(memory_format=torch.contiguous_format)
        return sd

    def load_state_dict(self, state_dict, strict=True, **kw):
        sd = {k: v for k, v in state_dict.items() if not k.endswith("num_batches_tracked")}   # backbone.py:59-63
        out = super().load_state_dict(sd, strict=strict, **kw)
        self.mark_dirty(full=True)
        return out

    def _apply(self, fn, recurse=True):
        probe = fn(torch.zeros(1, device=self.store.device))
        if probe.device != self.store.device:
            old = self.store.flat
            self.store.allocate(probe.device, src=old)
            rebind(self, self.store)
            self.body = ResNetBody(self.store, self.cfg)
            self.net = Net(self.store, self.cfg)
            self.seg = SegHead(self.store, self.cfg, self.net) if self.cfg.masks else None
            self._anchor = torch.zeros((), device=probe.device, requires_grad=True)
            self.seed_dev = self.seed_dev.to(probe.device)
            self._operand_version = getattr(self, "_operand_version", 0) + 1      # the optimizer's operand tables point at the old tensors
            self.mark_dirty(full=True)
        return self

    def init_from_pretrained_detr(self, state_dict):
        """models/reftr_transformer.py:137-146."""
        bb = {"img_backbone." + k.split(".", 1)[1]: v for k, v in state_dict.items() if k.split(".", 1)[0] == "backbone"}
        enc = {"vl_transformer.encoder." + k.split(".", 2)[2]: v for k, v in state_dict.items() if "transformer.encoder" in k}
        bb.update(enc)
        self.load_state_dict(bb, strict=False)

    def init_from_pretrained(self, state_dict):
        """RefTRSeg.init_from_pretrained (models/reftr_segmentation.py:66-74): non-strict load of a REC checkpoint."""
        missing, unexpected = self.load_state_dict(state_dict, strict=False)
        print("Unexpected keys: ", unexpected)
        print("Missing keys: ", missing)

    # ------------------------------------------------------------------ forward
    def forward(self, samples, _logits_only=False):
        """`_logits_only` (the captured training step's direct loss path, engine_vg.CapturedTrainStep._fwd_bwd): returns the
        pre-sigmoid box logits [NL, B, P, K, 4] without an autograd node and without the sigmoid of `pred_boxes`, which nothing in
        a training iteration reads (engine_vg.py:40-60 only consumes the criterion's losses); the caller owns the backward."""
        self._bb_ready = None
        H.mark("step start")
        if self._pre_update is not None:
            # the pending (deferred) AdamW pass over the main / mask / ResNet slices; when it wrote the bf16 operands itself
            # (rt_adamw_mat) nothing is dirty.  The BERT slice's pass opens the language branch (_forward_impl).
            if not self._pre_update[0]():
                self.mark_dirty()
        elif self._flush_pending is not None:
            self._flush_pending()
        self.refresh_operands()
        H.mark("AdamW (main slice) + operands done")
        pred_masks = None
        if _logits_only:
            assert self.seg is None
            with torch.no_grad():
                return self._forward_impl(samples)
        if self.seg is not None:
            assert "phrase" not in samples, "RefTRSeg is single-phrase (reftr_segmentation.py:101-103)"
            res = _RefTRSegFunction.apply(self._anchor, self, samples)
            logits, pred_masks = res[0], res[1]
            cem_loss = res[2] if self.cfg.cem else None
        else:
            logits = _RefTRFunction.apply(self._anchor, self, samples)      # [NL, B, P, K, 4]
        boxes = logits.sigmoid()
        phrase_mask = self._saved["phrase_mask"]
        out = {"pred_boxes": boxes[-1], "phrase_mask": phrase_mask, "pred_logits": logits}
        if pred_masks is not None:                                          # reftr_segmentation.py:147-148
            out["pred_masks"] = pred_masks
            out["mask_att"] = self._saved["mask_att"]
            if cem_loss is not None:                                        # reftr_segmentation.py:145-146
                out["cem_loss"] = cem_loss[0]
        if self.aux_loss:
            out["aux_outputs"] = [{"pred_boxes": b, "phrase_mask": phrase_mask} for b in boxes[:-1]]
        return out

    def _forward_impl(self, samples):
        cfg, net, st = self.cfg, self.net, self.store
        dev = st.device
        E = cfg.hidden
        img = samples["img"]
        if not isinstance(img, NestedTensor):
            img = nested_tensor_from_tensor_list(img)
        x, mask = img.decompose()
        x = x.to(dev, torch.float32).contiguous()
        mask_u8 = H.as_u8(mask.to(dev))
        B = x.shape[0]
        self._step += 1
        net.begin_step(self.training)
        H.set_seed_dev(self.seed_dev)
        if self.training:
            H.counter_add(self.seed_dev, 1)

        ids = samples["sentence"].to(dev).contiguous()
        # the sentence mask arrives as int64 (BERT's attention mask): its uint8 image is first needed by the language branch, which
        # converts it on ITS stream (one launch less on the main stream's head); readers behind the forward join find it there
        smask_src = samples["sentence_mask"].to(dev)
        smask_box = []

        def _smask():
            if not smask_box:
                smask_box.append(H.as_u8(smask_src))
            elif smask_box[0] is not smask_src and smask_box[0].is_cuda:
                smask_box[0].record_stream(torch.cuda.current_stream())
            return smask_box[0]
        Lq = ids.shape[1]
        # language branch (BERT) on the side stream, concurrently with the ResNet branch below
        # c5 geometry from the image size (stem 7x7/2, maxpool 3x3/2, three stride-2 stages): the positional / mask work below
        # only needs the padding mask, so it rides on the language side stream behind BERT
        h, w = x.shape[2], x.shape[3]
        for _ in range(4 if cfg.dilation else 5):                            # --dilation: layer4 keeps stride 16
            h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        HW = h * w
        assert Lq <= cfg.max_lang_seq                                       # models/reftr.py:81
        S = Lq + HW
        M = B * S
        vt = "vl_transformer."

        # the sequence buffers [B * S, E] (fp32 residual stream, its bf16 image, bf16 of x + pos): allocated in front of the fork so that
        # the language branch can write its rows of them (map_sentence) while the ResNet still runs
        x32 = torch.empty(M, E, dtype=torch.float32, device=dev)
        x16 = torch.empty(M, E, dtype=torch.bfloat16, device=dev)
        xp16 = torch.empty(M, E, dtype=torch.bfloat16, device=dev)
        lang_tail = {}

        def _lang_branch():
            H.mark("lang: branch starts")
            if self._pre_update is not None and self._pre_update[1]():
                net._refresh_kv_cat()          # operands written by the AdamW passes (this one and the main stream's, which is ordered in front of this branch)
            if self._lin_refresh_pending:
                net.refresh()
                self._lin_refresh_pending = False
            H.mark("lang: AdamW (BERT slice) + operands done")
            r = net.bert_fwd(ids, _smask())
            H.mark("lang: BERT forward done")
            res = r + _pos_work()
            if self._lang_tail:
                # what only BERT's outputs feed -- map_sentence into the language rows of the sequence, and with one phrase per image
                # the context mask and map_phrase -- stays on this stream instead of opening the main stream's chain behind the
                # forward join (5-6 launches; this branch has ~90 us of slack there).  Same order of the dropout sites as before.
                sq16, pl16, _, pos_, _ = res
                _, lang_tail["ms_ctx"] = net.mlp_fwd(sq16, "map_sentence.", y_f32=x32, y_bf16=x16, ypos_bf16=xp16, pos=pos_, rowmap=(Lq, S, 0))
                if "phrase" not in samples:
                    lang_tail["masks"] = H.context_mask(_smask())
                    cat16_ = torch.empty(B, 2 * E, dtype=torch.bfloat16, device=dev)
                    _, lang_tail["mp_ctx"] = net.mlp_fwd(pl16, "map_phrase.", y_bf16=cat16_.view(2 * B, E), want_f32=False, rowmap=(1, 2, 1))
                    lang_tail["cat16"] = cat16_
                H.mark("lang: map_sentence / map_phrase done")
            return res

        def _pos_work():
            pos = torch.empty(M, E, dtype=torch.float32, device=dev)
            kpm = torch.empty(B, S, dtype=torch.uint8, device=dev)
            kpm[:, :Lq] = (_smask() == 0)                                   # models/reftr.py:92
            H.rows_add(B * Lq, E, a_f32=st.P[vt + "lang_pos_embeddings.weight"], a_map=(Lq, 0, 0),
                       b_f32=st.P[vt + "token_type_embeddings.weight"], b_map=(-(B * Lq), 0, 0),
                       out_f32=pos, o_map=(Lq, S, 0))
            addv = torch.empty(E, dtype=torch.float32, device=dev)
            H.rows_add(1, E, a_f32=st.P[vt + "level_embed"], b_f32=st.P[vt + "token_type_embeddings.weight"],
                       b_map=(-1, 0, 1), out_f32=addv)
            H.mask_posenc(mask_u8, h, w, E, addv, kpm, Lq, pos, S, Lq)
            if cfg.pos_learned:
                # PositionEmbeddingLearned (position_encoding.py:74-84): pos[:, y, x] = [col_embed[x] | row_embed[y]], the same for
                # every image and independent of the padding mask; the image rows of `pos` (sine values above; the kernel is still
                # what downsamples the padding mask) are overwritten.  O(h * w * E) glue: a table lookup, no arithmetic to port.
                assert h <= 50 and w <= 50, "PositionEmbeddingLearned has 50 rows / columns (position_encoding.py:65-66)"
                col = st.P["img_backbone.1.col_embed.weight"][:w]; row = st.P["img_backbone.1.row_embed.weight"][:h]
                pe = torch.cat([col.unsqueeze(0).expand(h, w, -1), row.unsqueeze(1).expand(h, w, -1)], dim=-1).reshape(HW, E) + addv
                pos.view(B, S, E)[:, Lq:, :] = pe
            H.mark("lang: positional / mask work done")
            return (pos, kpm)
        # REFTR_STEM_FIRST=1: the language branch is forked BEHIND the frozen stem (rt_stem_pool: two 72-KB / 230-VGPR workgroups per
        # CU, which find no room beside the BERT slice's AdamW pass once that has filled the chip) instead of in front of it
        stem_first = 1 if self.body.fuse_stem else 0                        # (behind frozen layer1 as well, or in front of the stem: slower, r04al)
        lang_box = []
        def _fork_lang():
            lang_box.append(net.side.run(_lang_branch, ids, smask_src, mask_u8))
        if not stem_first:
            _fork_lang()
        feats, bb_saved = self.body.forward(x, ready=self._bb_ready, after_stem=_fork_lang if stem_first else None)
        seq16, pooled16, bctx, pos, kpm = lang_box[0]
        c5, (_, h5, w5) = feats[-1]
        assert (h5, w5) == (h, w)
        H.mark("ResNet forward done")
        net.side.join()
        H.mark("forward join (language branch in)")
        if "ms_ctx" in lang_tail:
            ms_ctx = lang_tail["ms_ctx"]
        else:
            _, ms_ctx = net.mlp_fwd(seq16, "map_sentence.", y_f32=x32, y_bf16=x16, ypos_bf16=xp16, pos=pos, rowmap=(Lq, S, 0))
        _, ip = net.lin_fwd("input_proj.0.0.", c5, out_bf16=False, out_f32=True)
        gn_stats = H.groupnorm_fwd(ip.view(B, HW, E), st.P["input_proj.0.1.weight"], st.P["input_proj.0.1.bias"], 32, 1e-5,
                                   y_f32=x32, y_bf16=x16, pos=pos, ypos_bf16=xp16, rows_per_img=S, row_off=Lq)

        pctx = None
        if "phrase" in samples:
            ph = samples["phrase"].to(dev)
            Pn, Lp = ph.shape[1], ph.shape[2]
            pm_u8 = H.as_u8(samples["phrase_mask"].to(dev))
            _, ph_pooled16, pctx = net.bert_fwd(ph.reshape(B * Pn, Lp).contiguous(), pm_u8.view(B * Pn, Lp))
            ctxmask, qmask = H.context_mask(_smask(), pm_u8, samples["phrase_pos_l"].to(dev).contiguous(),
                                            samples["phrase_pos_r"].to(dev).contiguous())
        else:
            Pn = 1
            ph_pooled16 = pooled16
            ctxmask, qmask = lang_tail["masks"] if "masks" in lang_tail else H.context_mask(_smask())
        N = B * Pn
        if "mp_ctx" in lang_tail:
            cat16, mp_ctx = lang_tail["cat16"], lang_tail["mp_ctx"]
            cat_rows = cat16.view(2 * N, E)
        else:
            cat16 = torch.empty(N, 2 * E, dtype=torch.bfloat16, device=dev)
            cat_rows = cat16.view(2 * N, E)
            _, mp_ctx = net.mlp_fwd(ph_pooled16, "map_phrase.", y_bf16=cat_rows, want_f32=False, rowmap=(1, 2, 1))

        src32 = x32                        # sequence buffer whose image rows hold input_proj + GroupNorm (img_src_proj)
        enc = []
        nxt = None
        for i in range(cfg.enc_layers):
            x32, x16, xp16, r, nxt = net.enc_layer_fwd(f"{vt}encoder.layers.{i}.", x32, x16, xp16, pos, kpm, B, S, qkv=nxt,
                                                       next_p=f"{vt}encoder.layers.{i + 1}." if i + 1 < cfg.enc_layers else None)
            enc.append(r)
        mem32, mem16, memp16 = x32, x16, xp16
        H.mark("encoder forward done")

        # ---- QueryEncoder (models/reftr_transformer.py:41-66)
        qe = "query_encoder."
        nq = cfg.n_q
        Nf = N                                                              # fused phrase rows
        qf = self._qfuse_ok(Lq, Pn)
        if qf:
            # round 5: the whole QueryEncoder + the query / query_pos split as ONE launch (rt_qenc_fwd), same saved tensors
            q = self._qenc_fwd_fused(mem16, mem32, ctxmask, cat16, B, S, Lq, Pn)
            cls16, lang16, kq, qs, vs, qw, c16, co = (q[k] for k in ("cls16", "lang16", "kq", "qs", "vs", "qw", "c16", "co"))
            cm, cr, fq_ctx = q["cmean"], q["crstd"], q["fq_ctx"]
            tgt32, tgt16, qpos, tgtq16 = q["tgt32"], q["tgt16"], q["qpos"], q["tgtq16"]
            N = N * nq
            if nq > 1:
                qmask = qmask.view(B, Pn, 1).expand(B, Pn, nq).reshape(B, Pn * nq).contiguous()      # :237-238
        else:
            cls16, lang16, kq, qs, vs, qw, c16, co, cm, cr, fq_ctx, tgt32, tgt16, qpos, tgtq16, qmask = self._qenc_fwd_chain(
                mem16, mem32, ctxmask, cat16, cat_rows, qmask, B, S, Lq, Pn)
            N = N * nq

        # ---- decoder (models/modeling/transformer.py:105-143) + shared norm on every layer's output
        T = Pn * cfg.n_q
        NL = cfg.dec_layers
        hs16 = torch.empty(NL * N, E, dtype=torch.bfloat16, device=dev)
        dec = []
        t3_all = torch.empty(NL * N, E, dtype=torch.float32, device=dev)      # every layer's output, for the shared norm
        t32, t16, tq16 = tgt32, tgt16, tgtq16
        kvs = net.dec_kv_all([f"{vt}decoder.layers.{i}." for i in range(NL)], mem16, memp16) if NL else []
        fold_sa = "phrase" not in samples and nq == 1
        coop = NL > 0 and net.dec_stack_coop_ok(N, T, S, NL, fold_sa)
        if coop:
            dec = net.dec_stack_fwd_coop([f"{vt}decoder.layers.{i}." for i in range(NL)], t32, t16, tq16, qpos, kvs, kpm, B, S, t3_all)
        for i in range(0 if coop else NL):
            t32, t16, tq16, r = net.dec_layer_fwd(f"{vt}decoder.layers.{i}.", t32, t16, tq16, qpos, mem16, memp16,
                                                  qmask, kpm, B, T, S, kv=kvs[i], t3_out=t3_all[i * N:(i + 1) * N],
                                                  fold_sa=fold_sa)
            dec.append(r)
        # decoder.norm on every layer's output (transformer.py:131-141, return_intermediate): ONE launch over the stack
        hm = hr = None
        head = None
        hf = self.__dict__.get("_head_fused")
        if hf is not None and NL and self.seg is None and self.aux_loss and self._qfuse_ok(Lq, Pn):
            # round 5, captured step with the direct loss path: decoder.norm + box head + box losses + d total / d logits + the
            # head's and the norm's backward-data as ONE launch (rt_head_loss); backward starts at the decoder
            head = self._head_loss_fused(hf, t3_all, hs16, qmask.view(B, T), NL, B, Pn, nq, N, invert=True)
            hm, hr, y1, y2, logits = head["hmean"], head["hrstd"], head["y1"], head["y2"], head["logits"]
        else:
            if NL:
                _, _, _, hm, hr = net.ln_fwd(t3_all, vt + "decoder.norm.", y_bf16=hs16, want_f32=False)
            y1, _ = net.lin_fwd("bbox_embed.layers.0.", hs16, act=RELU)
            y2, _ = net.lin_fwd("bbox_embed.layers.1.", y1, act=RELU)
            _, logits = net.lin_fwd("bbox_embed.layers.2.", y2, out_bf16=False, out_f32=True)
        hs_stats, t3s = (hm, hr), t3_all

        self._saved = dict(
            B=B, S=S, Lq=Lq, HW=HW, Pn=Pn, N=N, Nf=Nf, T=T, NL=NL, bb_saved=bb_saved, c5=c5, bctx=bctx, pctx=pctx, ms_ctx=ms_ctx,
            mp_ctx=mp_ctx, ip=ip, gn_stats=gn_stats, kpm=kpm, qmask=qmask, ctxmask=ctxmask, enc=enc, mem16=mem16,
            memp16=memp16, mem32=mem32, cls16=cls16, lang16=lang16, kq=kq, qs=qs, vs=vs, qw=qw, c16=c16, co=co,
            cst=(cm, cr), fq_ctx=fq_ctx, dec=dec, hs_stats=hs_stats, t3s=t3s, hs16=hs16, y1=y1, y2=y2, pooled16=pooled16,
            # (the fused head reads the query mask itself and nothing in a captured training iteration reads the bool mask)
            phrase_mask=(qmask == 0).view(B, T) if head is None else None, qmask_bt=qmask.view(B, T), memory=mem32, hw=(h, w), head=head)
        if self.seg is not None:            # RES head on the last decoder layer (reftr_segmentation.py:136-146)
            pad_u8 = kpm[:, Lq:].contiguous()
            pred_masks, mask_att, seg_sv = self.seg.forward(hs16[(NL - 1) * N:], mem16, mem32, src32, pad_u8, feats, B, S, Lq, h, w)
            self._saved.update(pred_masks=pred_masks, mask_att=mask_att, seg=seg_sv)
        H.set_seed_dev(None)
        H.mark("query encoder + decoder + head forward done")
        return logits.view(NL, B, Pn, cfg.n_q, 4)

    # ------------------------------------------------------------------ QueryEncoder forward: launched chain / one launch
    def _qfuse_ok(self, Lq, Pn):
        """rt_qenc_fwd / rt_qenc_bwd apply (hidden 256, <= 96 tokens, <= 16 phrases per image) and are switched on."""
        return (os.environ.get("REFTR_QFUSE", "1") != "0" and self.cfg.hidden == 256 and Lq <= 96 and Pn <= 16
                and self.net.small_wg is not None and self.net.ln_batch is not None)

    def _qenc_fwd_chain(self, mem16, mem32, ctxmask, cat16, cat_rows, qmask, B, S, Lq, Pn):
        cfg, net, st = self.cfg, self.net, self.store
        dev, E = st.device, cfg.hidden
        N = B * Pn
        qe = "query_encoder."
        cls16 = torch.empty(B, E, dtype=torch.bfloat16, device=dev)
        lang16 = torch.empty(B * Lq, E, dtype=torch.bfloat16, device=dev)
        H.rows_add(B, E, a_bf16=mem16, a_map=(1, S, 0), out_bf16=cls16)
        H.rows_add(B * Lq, E, a_bf16=mem16, a_map=(Lq, S, 0), out_bf16=lang16)
        _, kq = net.lin_fwd(qe + "linear1.", cls16, out_bf16=False, out_f32=True)
        _, qs = net.lin_fwd(qe + "linear2.", lang16, out_bf16=False, out_f32=True)
        _, vs = net.lin_fwd(qe + "linear3.", lang16, out_bf16=False, out_f32=True)
        qw, qc = H.qenc_attn_fwd(kq, qs.view(B, Lq, E), vs.view(B, Lq, E), ctxmask)
        c16 = torch.empty(N, E, dtype=torch.bfloat16, device=dev)
        H.rows_add(N, E, a_f32=qc.view(N, E), out_bf16=c16)
        _, co = net.lin_fwd(qe + "context_out.0.", c16, out_bf16=False, out_f32=True)
        cn32, _, _, cm, cr = net.ln_fwd(co, qe + "context_out.1.", want_bf16=False)
        H.rows_add(N, E, a_f32=cn32, b_f32=mem32, b_map=(-Pn, S, 0), out_bf16=cat_rows, o_map=(1, 2, 0))
        (f32, _, _, _, _), fq_ctx = net.mlp_fwd(cat16, qe + "fuse_encoder_query.", want_bf16=False)
        # phrase_queries = fused.repeat(1, 1, 1, 2) + query_embed.view(1, 1, n_q, -1), phrase-major (:60-64): query row
        # (b * Pn + ph) * n_q + q = fused[b, ph] + query_embed[q], first half -> tgt, second half -> query_pos
        nq = cfg.n_q
        N = N * nq                                                          # query rows
        tgt32 = torch.empty(N, E, dtype=torch.float32, device=dev); tgt16 = torch.empty(N, E, dtype=torch.bfloat16, device=dev)
        qpos = torch.empty(N, E, dtype=torch.float32, device=dev); tgtq16 = torch.empty(N, E, dtype=torch.bfloat16, device=dev)
        if nq == 1:
            emb = st.P[qe + "query_embed.weight"].view(2, E)               # rows [tgt part | query_pos part]
            H.rows_add(N, E, a_f32=f32, b_f32=emb, b_map=(-N, 0, 0), out_f32=tgt32, out_bf16=tgt16)
            H.rows_add(N, E, a_f32=f32, b_f32=emb, b_map=(-N, 0, 1), out_f32=qpos)
        else:
            emb2 = st.P[qe + "query_embed.weight"].view(nq, 2, E)
            emb_t, emb_p = emb2[:, 0].contiguous(), emb2[:, 1].contiguous()       # [n_q, E] each (n_q x 1 KB of glue)
            H.rows_add(N, E, a_f32=f32, a_map=(-nq, 1, 0), b_f32=emb_t, b_map=(nq, 0, 0), out_f32=tgt32, out_bf16=tgt16)
            H.rows_add(N, E, a_f32=f32, a_map=(-nq, 1, 0), b_f32=emb_p, b_map=(nq, 0, 0), out_f32=qpos)
            qmask = qmask.view(B, Pn, 1).expand(B, Pn, nq).reshape(B, Pn * nq).contiguous()      # :237-238
        H.rows_add(N, E, a_f32=tgt32, b_f32=qpos, out_bf16=tgtq16)

        return cls16, lang16, kq, qs, vs, qw, c16, co, cm, cr, fq_ctx, tgt32, tgt16, qpos, tgtq16, qmask

    def _qenc_fwd_fused(self, mem16, mem32, ctxmask, cat16, B, S, Lq, Pn):
        cfg, net, st = self.cfg, self.net, self.store
        dev, E, nq = st.device, cfg.hidden, cfg.n_q
        qe = "query_encoder."
        N = B * Pn
        f32 = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)       # noqa: E731
        b16 = lambda *sh: torch.empty(*sh, dtype=torch.bfloat16, device=dev)      # noqa: E731
        L1, L2, L3, Lc = (net.lins[qe + k] for k in ("linear1.", "linear2.", "linear3.", "context_out.0."))
        F0, F4 = net.lins[qe + "fuse_encoder_query.0."], net.lins[qe + "fuse_encoder_query.4."]
        dp, ds = net._drop(0.1)                                              # the site mlp_fwd would have drawn
        o = dict(cls16=b16(B, E), lang16=b16(B * Lq, E), kq=f32(B, E), qs=f32(B * Lq, E), vs=f32(B * Lq, E), qw=f32(B, Pn, Lq),
                 c16=b16(N, E), co=f32(N, E), cmean=f32(N), crstd=f32(N), t1=f32(N, E), m1=f32(N), r1=f32(N), a16=b16(N, E),
                 t2=f32(N, E), m2=f32(N), r2=f32(N), tgt32=f32(N * nq, E), tgt16=b16(N * nq, E), qpos=f32(N * nq, E),
                 tgtq16=b16(N * nq, E))
        H.qenc_fwd(B=B, S=S, L=Lq, P=Pn, nq=nq, E=E, eps=1e-5, drop_p=dp, drop_seed=ds,
                   mem16=mem16, mem32=mem32, ctx=ctxmask, cat16=cat16,
                   W1=L1.W, W2=L2.W, W3=L3.W, Wc=Lc.W, Wf0=F0.W, Wf4=F4.W, b1=L1.b32, b2=L2.b32, b3=L3.b32, bc=Lc.b32, bf0=F0.b32, bf4=F4.b32,
                   gc=st.P[qe + "context_out.1.weight"], betc=st.P[qe + "context_out.1.bias"],
                   g1=st.P[qe + "fuse_encoder_query.1.weight"], bet1=st.P[qe + "fuse_encoder_query.1.bias"],
                   g5=st.P[qe + "fuse_encoder_query.5.weight"], bet5=st.P[qe + "fuse_encoder_query.5.bias"],
                   qembed=st.P[qe + "query_embed.weight"], **o)
        o["fq_ctx"] = {"x16": cat16, "t1": o["t1"], "st1": (o["m1"], o["r1"], dp, ds), "a16": o["a16"], "t2": o["t2"],
                       "st2": (o["m2"], o["r2"])}
        return o

    def _head_ticket(self):
        """The persistent arrival counter of rt_head_loss (zero between launches)."""
        t = self.__dict__.get("_head_ticket_buf")
        if t is None:
            t = self.__dict__["_head_ticket_buf"] = torch.zeros(1, dtype=torch.int32, device=self.store.device)
        return t

    def _head_loss_fused(self, hf, t3_all, hs16, mask, NL, B, Pn, nq, N, invert=False):
        """rt_head_loss over the NL * N decoder-output rows; `hf` = (criterion, (boxes, offsets, num_boxes)) from the engine.
        `mask` [B, T]: the bool phrase mask (True = a real phrase), or -- invert -- the model's uint8 query mask (1 = ignore)."""
        from .criterion import _box_weights
        crit, (boxes, off, num_boxes) = hf
        net, st, E = self.net, self.store, self.cfg.hidden
        dev = st.device
        vt = "vl_transformer."
        f32 = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)       # noqa: E731
        b16 = lambda *sh: torch.empty(*sh, dtype=torch.bfloat16, device=dev)      # noqa: E731
        l0, l1, l2 = (net.lins[f"bbox_embed.layers.{i}."] for i in range(3))
        R = NL * N
        o = dict(hs16=hs16, y1=b16(R, E), y2=b16(R, E), hmean=f32(R), hrstd=f32(R), logits=f32(R, 4), losses=f32(NL, 2),
                 dlogits=f32(R, 4), dl16=b16(R, 4), dy2=b16(R, E), dy1=b16(R, E), dhs=f32(R, E), dnorm=f32(R, E), part_n=f32(NL, 2, E),
                 db2_part=f32(NL, 2, 4), total=f32(1))
        H.head_loss(NL=NL, B=B, P=Pn, K=nq, E=E, eps=1e-5, t3=t3_all,
                    gn=st.P[vt + "decoder.norm.weight"], betn=st.P[vt + "decoder.norm.bias"],
                    W0=l0.W, W1=l1.W, W2=l2.W, W0T=l0.WT, W1T=l1.WT, b0=l0.b32, b1=l1.b32, b2=l2.b32, w2_f32=l2.w32,
                    valid=H.as_u8(mask), invert_valid=invert,
                    targets=boxes, tgt_off=off, num_boxes=num_boxes, weights=_box_weights(crit, NL, t3_all.device),
                    ticket=self._head_ticket(), **o)     # the device object rt_box_loss's path keys its cache with
        return o

    def _qenc_bwd_chain(self, sv, ga, gb, dqpos, dmem, B, S, Lq, Pn, N):
        cfg, net, st = self.cfg, self.net, self.store
        dev, E = st.device, cfg.hidden
        qe = "query_encoder."
        nq, Nf = cfg.n_q, sv["Nf"]
        df = torch.empty(N, E, dtype=torch.float32, device=dev)
        if nq == 1:
            gqe = st.G[qe + "query_embed.weight"].view(2, E)
            H.colsum(ga, gqe[0]); H.colsum(dqpos, gqe[1])
        if gb is None:                  # trivial self-attention (one query per image): no gradient through t + query_pos
            H.rows_add(N, E, a_f32=ga, b_f32=dqpos, out_f32=df)
        else:
            if nq == 1:
                H.colsum(gb, gqe[0])
            H.rows_add(N, E, a_f32=ga, b_f32=gb, out_f32=df)
            H.rows_add(N, E, a_f32=dqpos, out_f32=df, accumulate=True)
        if nq > 1:
            # query row (phrase, q): d query_embed[q] = sum over phrases, d fused[phrase] = sum over q (rows of a few KB: glue)
            dt = (ga if gb is None else ga + gb).view(Nf, nq, E)
            g2 = st.G[qe + "query_embed.weight"].view(nq, 2, E)
            g2[:, 0] += dt.sum(0); g2[:, 1] += dqpos.view(Nf, nq, E).sum(0)
            df = df.view(Nf, nq, E).sum(1).contiguous()
        N = Nf                          # from here on: one row per phrase
        dcat = net.mlp_bwd(sv["fq_ctx"], df, qe + "fuse_encoder_query.")            # fp32 [N, 2E]
        dcat_rows = dcat.view(2 * N, E)
        _, dcob = net.ln_bwd(dcat_rows, sv["co"], qe + "context_out.1.", *sv["cst"], rowmap=(1, 2, 0), want_f32=False)
        _, dc = net.lin_bwd(qe + "context_out.0.", dcob, sv["c16"], out_bf16=False, out_f32=True)
        H.rows_add(N, E, a_f32=dcat_rows, a_map=(1, 2, 0), out_f32=dmem, accumulate=2, o_map=(-Pn, S, 0))
        dk, dqs, dvs = H.qenc_attn_bwd(sv["kq"], sv["qs"].view(B, Lq, E), sv["vs"].view(B, Lq, E), sv["qw"], dc.view(B, Pn, E))
        dk16 = torch.empty(B, E, dtype=torch.bfloat16, device=dev)
        dqs16 = torch.empty(B * Lq, E, dtype=torch.bfloat16, device=dev); dvs16 = torch.empty_like(dqs16)
        H.rows_add(B, E, a_f32=dk, out_bf16=dk16)
        H.rows_add(B * Lq, E, a_f32=dqs.view(B * Lq, E), out_bf16=dqs16)
        H.rows_add(B * Lq, E, a_f32=dvs.view(B * Lq, E), out_bf16=dvs16)
        _, dcls = net.lin_bwd(qe + "linear1.", dk16, sv["cls16"], out_bf16=False, out_f32=True)
        H.rows_add(B, E, a_f32=dcls, out_f32=dmem, accumulate=True, o_map=(1, S, 0))
        _, dla = net.lin_bwd(qe + "linear2.", dqs16, sv["lang16"], out_bf16=False, out_f32=True)
        _, dlang = net.lin_bwd(qe + "linear3.", dvs16, sv["lang16"], res_f32=dla, out_bf16=False, out_f32=True)
        H.rows_add(B * Lq, E, a_f32=dlang, out_f32=dmem, accumulate=True, o_map=(Lq, S, 0))
        return dcat, dcat_rows

    def _qenc_bwd_fused(self, sv, ga, gb, dqpos, dmem, B, S, Lq, Pn):
        cfg, net, st = self.cfg, self.net, self.store
        dev, E = st.device, cfg.hidden
        qe = "query_encoder."
        N = B * Pn
        f32 = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)       # noqa: E731
        b16 = lambda *sh: torch.empty(*sh, dtype=torch.bfloat16, device=dev)      # noqa: E731
        L1, L2, L3, Lc = (net.lins[qe + k] for k in ("linear1.", "linear2.", "linear3.", "context_out.0."))
        F0, F4 = net.lins[qe + "fuse_encoder_query.0."], net.lins[qe + "fuse_encoder_query.4."]
        fq = sv["fq_ctx"]
        m1, r1, dp, ds = fq["st1"]
        m2, r2 = fq["st2"]
        cm, cr = sv["cst"]
        parts = f32(3, B, 2, E)
        o = dict(dt2b=b16(N, E), dt1b=b16(N, E), dcob=b16(N, E), dk16=b16(B, E), dqs16=b16(B * Lq, E), dvs16=b16(B * Lq, E),
                 da=f32(N, E), dcat=f32(N, 2 * E), dc=f32(N, E))
        H.qenc_bwd(B=B, S=S, L=Lq, P=Pn, E=E, drop_p=dp, drop_seed=ds, ga=ga, gb=gb, dqpos=dqpos,
                   t2=fq["t2"], m2=m2, r2=r2, g5=st.P[qe + "fuse_encoder_query.5.weight"], bet5=st.P[qe + "fuse_encoder_query.5.bias"],
                   t1=fq["t1"], m1=m1, r1=r1, g1=st.P[qe + "fuse_encoder_query.1.weight"], bet1=st.P[qe + "fuse_encoder_query.1.bias"],
                   co=sv["co"], cmean=cm, crstd=cr, gc=st.P[qe + "context_out.1.weight"], betc=st.P[qe + "context_out.1.bias"],
                   kq=sv["kq"], qs=sv["qs"], vs=sv["vs"], qw=sv["qw"],
                   Wf4T=F4.WT, Wf0T=F0.WT, WcT=Lc.WT, W1T=L1.WT, W2T=L2.WT, W3T=L3.WT,
                   dmem=dmem, dqembed=st.G[qe + "query_embed.weight"], part5=parts[0], part1=parts[1], partc=parts[2], **o)
        net._wgrad_only(qe + "fuse_encoder_query.4.", o["dt2b"], fq["a16"])
        net._wgrad_only(qe + "fuse_encoder_query.0.", o["dt1b"], fq["x16"])
        net._wgrad_only(qe + "context_out.0.", o["dcob"], sv["c16"])
        net._wgrad_only(qe + "linear1.", o["dk16"], sv["cls16"])
        net._wgrad_only(qe + "linear2.", o["dqs16"], sv["lang16"])
        net._wgrad_only(qe + "linear3.", o["dvs16"], sv["lang16"])
        for i, pfx in enumerate(("fuse_encoder_query.5.", "fuse_encoder_query.1.", "context_out.1.")):
            net.ln_batch.jobs.append(H.LnPgJob(H._p(parts[i]), H._p(st.G[qe + pfx + "weight"]), H._p(st.G[qe + pfx + "bias"]), B, E))
        net.ln_batch.keep.append(parts)
        return o["dcat"]

    # ------------------------------------------------------------------ backward
    BOUNDARIES_SERIAL = ("main", "bert_hi", "bert_mid", "bert", "layer4")
    BOUNDARIES_INTERLEAVE = ("main", "pair4", "pair3")     # pair4 = BERT layers 11-8 + ResNet layer4, pair3 = BERT 7-4 + layer3

    @property
    def BOUNDARIES(self):
        return self.BOUNDARIES_SERIAL if self.dp_schedule == "serial" else self.BOUNDARIES_INTERLEAVE

    def active_boundaries(self):
        """The boundaries this model's backward actually passes (serial schedule: BERT is cut in thirds only from 3 layers up)."""
        return tuple(b for b in self.BOUNDARIES
                     if (self.cfg.bert.layers >= 3 or b not in ("bert_hi", "bert_mid"))
                     and (self.cfg.train_backbone or b != "layer4"))       # frozen ResNet (--lr_backbone 0): its backward never runs

    def bert_cuts(self):
        """{BERT layer index: boundary reached right after that layer's backward} of the interleaved schedule
        (REFTR_DDP_BERT_CUTS; compute cost measured through single-rank RCCL, profiles/r03_ddp_single_rank.txt):
          "3"  thirds (round 2): layers 11-8 + pooler beside ResNet layer4, 7-4 beside layer3, 3-0 + embeddings beside layer2 --
               cheapest compute (8.21 ms), but the last third (107 MB in bf16) becomes final together with layer2 and is
               exchanged exposed;
          "2"  halves (round 3 default): layers 11-6 + pooler beside layer4, layers 5-0 + embeddings beside layer3, so that every
               BERT byte is on the wire before layer2's backward starts and only layer2-3's slice (8.6 MB in bf16) is exposed at
               the end; +0.13 ms of compute (the first half outlasts layer4's backward);
          "1"  one cut at two thirds (11-8 | 7-0 + embeddings): +0.17 ms."""
        nl = self.cfg.bert.layers
        if nl < 3:
            return {}
        mode = os.environ.get("REFTR_DDP_BERT_CUTS", "2")
        if mode == "3":
            return {(2 * nl) // 3: "pair4", nl // 3: "pair3"}
        return {(nl // 2 if mode == "2" else (2 * nl) // 3): "pair4"}   # the walk's second leg runs to the end of BERT: "pair3"

    @property
    def dp_mode(self):
        return bool(self._stops) or any(self._phase_hooks.values())

    def _backward_impl(self, dlogits, dmasks=None, dcem=None):
        self._norm_split = None
        if self._bwd_gen is not None:
            # the previous backward stopped at a data-parallel boundary and was never resumed (an exception in the loop): its
            # half-written gradient set is void.  A backward armed for THIS step by zero_grad(fast=True) keeps its state (the
            # written set was reset when it was armed); anything else was fully cleared by the caller.
            self._bwd_gen.close()
            self._bwd_gen = None
        self._bwd_gen = self._backward_phases(dlogits, dmasks, dcem)
        self.continue_backward()

    def continue_backward(self):
        """Runs backward up to the next boundary listed in `_stops` (returns its name) or to the end (returns None);
        the hooks of every boundary passed on the way are called."""
        gen = self._bwd_gen
        if gen is None:
            return None
        for name in gen:
            # data parallel: the slices that are final at this boundary are about to be exchanged -- their registered matrices
            # that this step never wrote are cleared NOW (ParamStore.finish_overwrite_range), not at the end of backward
            sb = self._slice_bounds
            if sb is not None and name in sb:
                self.store.finish_overwrite_range(sb[name] if isinstance(sb[name], list) else [sb[name]])
            for hook in self._phase_hooks.get(name, ()):
                hook()
            if name in self._stops:
                return name
        self._bwd_gen = None
        H.set_seed_dev(None)
        self.store.finish_overwrite()
        if self._norm_split is not None:         # a matrix of the BERT slice cleared AFTER its squared norm was taken: void
            b0, b1, _ = self._norm_split
            if any(b0 <= off < b1 for off, _n in self.store.last_stale):
                self._norm_split = None
        for hook in self._post_backward_hooks:
            hook()
        return None

    def finish_backward(self):
        """Runs whatever is left of a backward that stopped at a boundary."""
        while self.continue_backward() is not None:
            pass

    def _backward_phases(self, dlogits, dmasks=None, dcem=None):
        cfg, net, st, sv = self.cfg, self.net, self.store, self._saved
        E = cfg.hidden
        dev = st.device
        H.set_seed_dev(self.seed_dev)
        B, S, Lq, HW, Pn, N, T, NL = (sv[k] for k in ("B", "S", "Lq", "HW", "Pn", "N", "T", "NL"))
        vt, qe = "vl_transformer.", "query_encoder."
        M = B * S
        f32z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)     # noqa: E731

        H.mark("loss done (backward starts)")
        head = sv.get("head")
        l2 = net.lins["bbox_embed.layers.2."]
        if head is not None:
            # rt_head_loss ran the head's backward-data with the forward: what is left are the weight gradients (the last Linear's
            # bias gradient was accumulated in the launch) and the norm's parameter gradients, all off the chain's data path
            assert dlogits is None and dmasks is None
            # (the last Linear's [4, E] weight gradient rides with the grouped launches too: inline it was two launches on the loss ->
            # decoder-backward chain)
            net.big_wg.add(head["dl16"], sv["y2"], l2.gw, None, overwrite=st.claim(l2.gw))
            net._wgrad_only("bbox_embed.layers.1.", head["dy2"], sv["y1"])
            net._wgrad_only("bbox_embed.layers.0.", head["dy1"], sv["hs16"])
            net.ln_batch.jobs.append(H.LnPgJob(H._p(head["part_n"]), H._p(st.G[vt + "decoder.norm.weight"]),
                                               H._p(st.G[vt + "decoder.norm.bias"]), NL, E))
            # the last Linear's bias gradient: the launch ran BEFORE the gradient clear (it sits in the forward), so it left per-layer
            # partial sums and the grouped reduction adds them to the bias gradient like a LayerNorm's d gamma
            net.ln_batch.jobs.append(H.LnPgJob(H._p(head["db2_part"]), H._p(l2.gb), None, NL, 4))
            net.ln_batch.keep.append((head["part_n"], head["db2_part"]))
            dhs = None
        else:
            # ---- bbox head (backbone.py:26-38)
            dl = dlogits.reshape(NL * N, 4)
            dl16 = torch.empty(NL * N, 4, dtype=torch.bfloat16, device=dev)
            H.rows_add(NL * N, 4, a_f32=dl, out_bf16=dl16)
            H.linear_wgrad(dl16, sv["y2"], l2.gw, overwrite=st.claim(l2.gw))
            H.colsum(dl, l2.gb)
            dy2 = H.small_dgrad(dl, l2.w32, gate=sv["y2"])
            dy1, _ = net.lin_bwd("bbox_embed.layers.1.", dy2, sv["y1"], gate=sv["y1"])
            _, dhs = net.lin_bwd("bbox_embed.layers.0.", dy1, sv["hs16"], out_bf16=False, out_f32=True)
        zz = f32z(2 * M + N, E)          # one clear for the three fp32 accumulators of the decoder backward
        dmem, dmemp, dqpos = zz[:M], zz[M:2 * M], zz[2 * M:]

        # ---- RES head (its gradients enter the last decoder output, the encoder memory, input_proj and the ResNet)
        seg_dsrc, seg_extra = None, None
        if dmasks is not None:
            d_hs, seg_dsrc, seg_extra = self.seg.backward(sv["seg"], dmasks, dmem, dcem)
            dhs[(NL - 1) * N:] += d_hs

        # ---- decoder
        ga = gb = None
        hm, hr = sv["hs_stats"]
        if head is not None:
            dnorm_all = head["dnorm"]
        else:
            dnorm_all, _ = net.ln_bwd(dhs, sv["t3s"], vt + "decoder.norm.", hm, hr, want_bf16=False)     # all layers, one launch
        coop_bwd = (NL > 0 and net.dec_coop_bwd and net.ln_batch is not None and net.small_wg is not None
                    and all(r.get("coop") for r in sv["dec"]))
        if coop_bwd:                    # the whole stack's backward chain as one cooperative launch (rt_decoder_bwd)
            ga = net.dec_stack_bwd_coop([f"{vt}decoder.layers.{i}." for i in range(NL)], sv["dec"], dnorm_all, sv["mem16"],
                                        sv["memp16"], sv["kpm"], B, S, dmem, dmemp, dqpos)
        for i in (() if coop_bwd else reversed(range(NL))):
            dnorm = dnorm_all[i * N:(i + 1) * N]
            extra = None
            if ga is not None and gb is None:
                extra = ga
            elif ga is not None:
                extra = torch.empty(N, E, dtype=torch.float32, device=dev)
                H.rows_add(N, E, a_f32=ga, b_f32=gb, out_f32=extra)
            ga, gb = net.dec_layer_bwd(f"{vt}decoder.layers.{i}.", sv["dec"][i], dnorm, extra, sv["mem16"], sv["memp16"],
                                       sv["qmask"], sv["kpm"], B, T, S, dmem, dmemp, dqpos)

        H.mark("head + decoder backward done")
        # ---- QueryEncoder backward
        nq, Nf = cfg.n_q, sv["Nf"]
        if nq == 1 and self._qfuse_ok(Lq, Pn):
            # round 5: the backward-data chain as ONE launch (rt_qenc_bwd); the six Linear weight gradients and the three LayerNorms'
            # parameter gradients are queued for the grouped launches as before
            dcat = self._qenc_bwd_fused(sv, ga, gb, dqpos, dmem, B, S, Lq, Pn)
            N = Nf
            dcat_rows = dcat.view(2 * N, E)
        else:
            dcat, dcat_rows = self._qenc_bwd_chain(sv, ga, gb, dqpos, dmem, B, S, Lq, Pn, N)
            N = Nf
        # map_phrase: gradient rows 2r+1 of dcat; its input is BERT's pooled output (tanh) -> fold tanh'
        def _map_phrase_bwd():
            return net.mlp_bwd(sv["mp_ctx"], dcat_rows, "map_phrase.", dy_rowmap=(1, 2, 1), dx_f32=False,
                               dtanh=sv["pooled16"] if sv["pctx"] is None else sv["pctx"]["pooled"])
        # Single process with the language stream on: the backward of map_phrase and of map_sentence only feed BERT's backward, so
        # they open THAT branch (language stream) instead of sitting in the main stream's chain (REFTR_LANG_TAIL, as in forward)
        lang_head = self._lang_tail and net.side.enabled and not self.dp_mode
        phrase_on_lang = lang_head
        dpool = None if phrase_on_lang else _map_phrase_bwd()

        # (the decoder / query-encoder / map_phrase / head weight gradients stay queued: they ride with the encoder's group on the
        # language stream below -- launched here, or beside the encoder's chain, they cost the chain more than they save:
        # profiles/r04bd_wgrad_merge_ab.txt, r03_side_stream_probes.txt)
        H.mark("query encoder backward done")
        # ---- encoder
        H.rows_add(M, E, a_f32=dmemp, out_f32=dmem, accumulate=True)      # K-side input of every cross-attention = memory + pos
        dpos = dmemp
        dmem_dbg = dmem.clone() if getattr(self, "_debug", False) else None
        dxa, dxb = dmem, None
        for i in reversed(range(cfg.enc_layers)):
            dxa, dxb = net.enc_layer_bwd(f"{vt}encoder.layers.{i}.", sv["enc"][i], dxa, dxb, sv["kpm"], B, S, dpos)
        H.pos_grad(dpos, st.G[vt + "lang_pos_embeddings.weight"], st.G[vt + "token_type_embeddings.weight"],
                   st.G[vt + "level_embed"], B, S, Lq)
        if cfg.pos_learned:             # d col_embed[x] = sum over images and rows, d row_embed[y] = sum over images and columns
            h5, w5 = sv["hw"]
            gi = dpos.view(B, S, E)[:, Lq:, :].reshape(B, h5, w5, E).sum(0)
            st.G["img_backbone.1.col_embed.weight"][:w5] += gi[..., :E // 2].sum(0)
            st.G["img_backbone.1.row_embed.weight"][:h5] += gi[..., E // 2:].sum(1)

        H.mark("encoder backward done")
        # ---- sequence inputs: map_sentence (language rows) and input_proj + GroupNorm (image rows)
        def _map_sentence_bwd():
            return net.mlp_bwd(sv["ms_ctx"], dxa, "map_sentence.", dy_rowmap=(Lq, S, 0), dy2=dxb)
        d_seq = None if lang_head else _map_sentence_bwd()
        if seg_dsrc is not None:
            H.rows_add(B * HW, E, a_f32=seg_dsrc, out_f32=dxa, accumulate=True, o_map=(HW, S, Lq))
        _, dip16 = H.groupnorm_bwd(dxa, sv["ip"].view(B, HW, E), st.P["input_proj.0.1.weight"], sv["gn_stats"],
                                   st.G["input_proj.0.1.weight"], st.G["input_proj.0.1.bias"], 32, 1e-5, dy2=dxb,
                                   rows_per_img=S, row_off=Lq)
        # input_proj's data / weight gradient (main group) -- in front of the BERT branch so that the main group is final first
        if cfg.train_backbone:
            g_c5, _ = net.lin_bwd("input_proj.0.0.", dip16.view(B * HW, E), sv["c5"], gate=sv["c5"])
        else:                                   # frozen ResNet (--lr_backbone 0): input_proj's weight gradient only
            net.lin_bwd("input_proj.0.0.", dip16.view(B * HW, E), sv["c5"], need_dx=False)
            g_c5 = None
        if getattr(self, "_debug", False):
            self._dbg = dict(dlogits=dlogits.clone(), dhs=dhs.clone(), dmem=dmem_dbg, g_c5=None if g_c5 is None else g_c5.clone())
        net.wg.flush()               # transformer weight gradients queued so far -> their own stream, from here
        dp = self.dp_mode
        # ---- BERT backward (sentence pass; phrase pass for multi-phrase inputs).  Single GPU: on the side stream,
        # concurrently with the ResNet backward.  Data parallel: on the main stream BEFORE the ResNet backward, so that every
        # non-ResNet gradient (> 80 % of the bytes) is exchanged under the ResNet backward; the exchange of a slice starts as
        # soon as it is final: the main group before BERT, BERT in thirds (layers 11-8 + pooler | 7-4 | 3-0 + embeddings).
        if dp and self.dp_schedule == "interleave":
            # [main slice final] | BERT 11-8 (language stream) || ResNet layer4 | BERT 7-4 || layer3 | BERT 3-0 + embeddings || layer2
            net.flush_wgrads()
            net.side.join(); net.wg.join()
            yield "main"
            nl = cfg.bert.layers
            cuts = self.bert_cuts()
            if sv["pctx"] is None:
                passes = [(sv["bctx"], d_seq, dpool)]
            else:
                passes = [(sv["bctx"], d_seq, None), (sv["pctx"], None, dpool)]

            def _bert_thirds():
                for n, (ctx, a, b_) in enumerate(passes):
                    last = n == len(passes) - 1          # a layer's gradient is final after the LAST pass through it
                    for layer in net.bert_bwd_layers(ctx, a, b_, stops=tuple(cuts) if last else ()):
                        yield cuts[layer]
            bg = _bert_thirds()
            rg = self.body.backward_stages(sv["bb_saved"], g_c5, seg_extra) if cfg.train_backbone else iter(())

            def _advance_bert():
                for _ in bg:
                    return                               # stopped at a cut: this third's gradients are launched
            for name in ("pair4", "pair3", None):
                net.side.run(_advance_bert, d_seq, dpool)
                for _ in rg:
                    break                                # one ResNet stage (4, 3, then 2 + what is left)
                if name is None:
                    for _ in rg:
                        pass
                net.flush_wgrads()
                net.side.join()
                net.wg.join()
                if name is not None:
                    yield name
            return
        if dp:
            net.flush_wgrads()       # encoder / map_sentence / input_proj weight gradients
            net.side.join(); net.wg.join()
            yield "main"
            nl = cfg.bert.layers
            cuts = {(2 * nl) // 3: "bert_hi", nl // 3: "bert_mid"} if nl >= 3 else {}
            if sv["pctx"] is None:
                passes = [(sv["bctx"], d_seq, dpool)]
            else:
                passes = [(sv["bctx"], d_seq, None), (sv["pctx"], None, dpool)]
            for n, (ctx, a, b_) in enumerate(passes):
                last = n == len(passes) - 1          # a layer's gradient is final after the LAST pass through it
                for layer in net.bert_bwd_layers(ctx, a, b_, stops=tuple(cuts) if last else ()):
                    net.wg.flush(); net.wg.join()
                    yield cuts[layer]
            net.wg.flush(); net.wg.join()
            yield "bert"
        else:
            def _bert_bwd():
                H.mark("lang: BERT backward starts")
                d_seq_ = _map_sentence_bwd() if lang_head else d_seq          # (their weight gradients: queued, launched with BERT's)
                dpool_ = _map_phrase_bwd() if phrase_on_lang else dpool
                if sv["pctx"] is None:
                    net.bert_bwd(sv["bctx"], d_seq_, dpool_)
                else:
                    net.bert_bwd(sv["bctx"], d_seq_, None)
                    net.bert_bwd(sv["pctx"], None, dpool_)
                net.wg.flush()           # BERT weight gradients: queued, launched behind the BERT data chain
                H.mark("lang: BERT backward (data + weight gradients) launched")
                if self._norm_side and not net.wg.enabled:      # (REFTR_STREAMS=3: the weight gradients run on their own stream -- not ordered in front of this one)
                    # Every gradient of the BERT slice (63 % of the buffer) is final here, ~1.5 ms before the ResNet backward ends:
                    # its share of the clip norm is reduced on this stream now; the optimizer adds the rest (optim.finish_step).
                    net.flush_wgrads()   # the slice's queued LayerNorm-parameter / grouped weight gradients (nothing else is queued)
                    b0, b1 = st.group_range[L.GROUP_BERT]
                    H.sqnorm(st.flat_g[b0:b1], self._sq_bert)
                    self._norm_split = (b0, b1, self._sq_bert)
                    H.mark("lang: BERT slice's squared norm done")
            H.mark("input_proj backward done (ResNet backward starts)")
            net.flush_wgrads_side()      # every transformer-side weight gradient queued so far: language stream, in front of the BERT branch
            net.side.run(_bert_bwd, d_seq, dpool, dxa, dxb, dcat_rows)
        # ---- ResNet body (its gradients are the last to become final); data parallel: layer4's slice (64 % of the ResNet
        # bytes) is final -- and exchanged -- before layer3 / layer2 run
        for stage in (self.body.backward_stages(sv["bb_saved"], g_c5, seg_extra) if cfg.train_backbone else ()):
            H.mark(f"ResNet backward: layer{stage} launched")
            if dp and stage == 4:
                net.flush_wgrads(); self.body.wgs.join()
                yield "layer4"
        net.flush_wgrads()
        H.mark("ResNet backward done")
        net.side.join()
        net.wg.join()
        H.mark("backward join (language branch in)")


def build_config(args):
    if int(getattr(args, "num_feature_levels", 1)) != 1:
        raise NotImplementedError("num_feature_levels > 1: the reference's own forward raises for every such value "
                                  "(models/reftr_transformer.py:171-175 feeds the 1024-channel map to the 512-channel input_proj[0] "
                                  "of :100-108) and every reference config uses 1, so there is no behaviour to match")
    # Flags the reference's builders read and this build does not implement must fail loudly -- a drop-in that silently
    # builds a different model is worse than none.  (--freeze_bert, --freeze_backbone and --freeze_reftr are parsed by
    # main_vg.py but have NO effect in the reference either: freeze_lang_backbone is stored and never used,
    # reftr_transformer.py:128,152-157; freeze_backbone is never read; build_reftr_seg hard-codes freeze_reftr=False,
    # reftr_segmentation.py:375.  They are accepted and ignored here for the same behaviour.)
    # (--dilation with --masks: built since round 6 -- the RES head takes its geometry from the feature maps it is given;
    # tests/golden/seg_dilation.npz, tests/test_seg_gpu.py)
    pe = getattr(args, "position_embedding", "sine")
    if pe not in ("v2", "sine"):
        raise ValueError(f"not supported {pe}")                     # as position_encoding.py:95
    if bool(getattr(args, "masks", False)) and getattr(args, "ablation", "none") == "cem_loss" and int(args.hidden_dim) != 256:
        raise NotImplementedError("--ablation cem_loss with hidden_dim != 256: the CEM kernels (rt_cem_fwd / rt_cem_bwd) are built for "
                                  "hidden_dim // 16 == 16 channels (every reference config uses hidden_dim 256)")
    backbone = str(getattr(args, "backbone", "resnet50"))
    if backbone not in ("resnet50", "resnet101"):
        raise NotImplementedError(f"--backbone {backbone}: the reference takes any torchvision ResNet name (models/modeling/backbone.py:"
                                  "119); this build has the bottleneck stacks of resnet50 (3, 4, 6, 3) and resnet101 (3, 4, 23, 3) only")
    bert_name = str(getattr(args, "bert_model", "bert-base-uncased"))
    if "base" not in bert_name.split("-"):
        raise NotImplementedError(f"--bert_model {bert_name}: the reference reads hidden_size from the checkpoint's config "
                                  "(models/reftr_transformer.py:84,315-318); this build has the base geometry (12 layers x 768, 12 heads) "
                                  "of bert-base-* / roberta-base only")
    # models/reftr_transformer.py:315-318: RobertaModel when args.bert_model starts with 'roberta', BertModel otherwise
    bc = L.roberta_config() if str(getattr(args, "bert_model", "bert-base-uncased")).split("-")[0] == "roberta" else L.BertConfig()
    layers = (3, 4, 23, 3) if getattr(args, "backbone", "resnet50") == "resnet101" else (3, 4, 6, 3)
    for k in ("bert_layers",):
        if hasattr(args, k):
            bc.layers = getattr(args, k)
    return L.ModelConfig(hidden=args.hidden_dim, nheads=args.nheads, enc_layers=args.enc_layers,
                         dec_layers=0 if getattr(args, "no_decoder", False) else args.dec_layers,
                         ffn=args.dim_feedforward, dropout=args.dropout, max_lang_seq=args.max_lang_seq,
                         n_q=args.num_queries_per_phrase, aux_loss=args.aux_loss, resnet_layers=layers, bert=bc,
                         masks=bool(getattr(args, "masks", False)),
                         # lr_backbone <= 0 freezes the whole ResNet (train_backbone = False, models/modeling/backbone.py:87-89,150):
                         # its parameters leave the optimizer and the clip norm, its backward is not run
                         train_backbone=float(getattr(args, "lr_backbone", 1e-5)) > 0,
                         dilation=bool(getattr(args, "dilation", False)),            # backbone.py:117-125 (DC5)
                         pos_learned=pe in ("v3", "learned"),          # position_encoding.py:91-92
                         cem=bool(getattr(args, "masks", False)) and getattr(args, "ablation", "none") == "cem_loss")


def build_reftr(args):
    """Same role as models/reftr_transformer.py:307-347: returns (model, criterion, postprocessors)."""
    from .criterion import CriterionVGMultiPhrase
    from .post_process import PostProcessVGMultiPhrase
    device = torch.device(args.device)
    cfg = build_config(args)
    model = RefTR(cfg, device=device, aux_loss=args.aux_loss)
    weight_dict = {"loss_giou": args.giou_loss_coef, "loss_bbox": args.bbox_loss_coef}
    if args.aux_loss:
        aux = {}
        for i in range(cfg.dec_layers - 1):
            aux.update({k + f"_{i}": v for k, v in weight_dict.items()})
        aux.update({k + "_enc": v for k, v in weight_dict.items()})
        weight_dict.update(aux)
    criterion = CriterionVGMultiPhrase(weight_dict, losses=["boxes"])
    postprocessors = {"bbox": PostProcessVGMultiPhrase()}
    criterion.to(device)
    return model, criterion, postprocessors


def build_reftr_seg(args):
    """models/reftr_segmentation.py:343-391: RefTRSeg + CriterionVGOnePhraseSeg + {'bbox', 'segm'} post-processors."""
    from .criterion import CriterionVGOnePhraseSeg
    from .post_process import PostProcessSegm, PostProcessVGMultiPhrase
    if args.reftr_type != "transformer_single_phrase":
        raise NotImplementedError                                   # as the reference (:389-390)
    device = torch.device(args.device)
    cfg = build_config(args)
    assert cfg.masks
    model = RefTR(cfg, device=device, aux_loss=False)
    weight_dict = {"loss_giou": args.giou_loss_coef, "loss_bbox": args.bbox_loss_coef,
                   "loss_dice": args.dice_loss_coef, "loss_mask": args.mask_loss_coef, "loss_cem": 1.0}
    if args.aux_loss:
        aux = {}
        for i in range(cfg.dec_layers - 1):
            aux.update({k + f"_{i}": v for k, v in weight_dict.items()})
        aux.update({k + "_enc": v for k, v in weight_dict.items()})
        weight_dict.update(aux)
    criterion = CriterionVGOnePhraseSeg(weight_dict, losses=["masks", "boxes"])
    postprocessors = {"bbox": PostProcessVGMultiPhrase(), "segm": PostProcessSegm()}
    criterion.to(device)
    return model, criterion, postprocessors
