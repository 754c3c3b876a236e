"""RES head of RefTRSeg on the MI355X kernel library: MHAttentionMap + MaskHeadSmallConv forward and hand-written
backward (reference: models/reftr_segmentation.py:151-280).

Every convolution is an rt_conv_gemm / rt_conv_wgrad launch over NHWC rows whose channel counts are padded to multiples
of 64 (models/layout.py:phys_dims — the padding is zero in weights, biases and activations, so it contributes nothing);
GroupNorm(8)+ReLU, the FPN nearest-upsample+add, the joint-softmax attention map and the concat are rt_seg.hip kernels.
Layout of the head's input rows X0[b*hw + p] = [input_proj(+GN) 256 | encoder memory 256 | attention map 8 | 0 x 56].
"""
import torch

from .. import hip as H
from . import layout as L


class _Conv:
    __slots__ = ("name", "cin", "cout", "k", "cip", "cop", "W", "WT", "w32", "gw", "b32", "gb")


class SegHead:
    def __init__(self, store, cfg, net):
        self.store, self.cfg, self.net = store, cfg, net
        dev = store.device
        self.convs = {}
        for name, ci, co, k in L.seg_convs(cfg):
            c = _Conv()
            c.name, c.cin, c.cout, c.k, c.cip, c.cop = name, ci, co, k, L.pad64(ci), L.pad64(co)
            c.w32 = store.phys(name + ".weight"); c.gw = store.phys(name + ".weight", grad=True)
            c.b32 = store.phys(name + ".bias"); c.gb = store.phys(name + ".bias", grad=True)
            c.W = torch.empty(c.cop, k * k, c.cip, dtype=torch.bfloat16, device=dev)
            c.WT = torch.empty(c.cip, k * k, c.cop, dtype=torch.bfloat16, device=dev)
            self.convs[name.split(".")[-1]] = c
        self._prep = None

    def refresh(self):
        if self._prep is None:
            self._prep = H.WeightPrepBatch(self.store.device)
            for c in self.convs.values():
                self._prep.add(c.w32, c.cop, c.k * c.k, c.cip, dst=c.W, dst_t=c.WT)
        self._prep.run()

    # ------------------------------------------------------------------ helpers
    def _conv(self, c, x, B, Hh, Ww):
        geom = (B, Hh, Ww, c.cip, Hh, Ww, c.cop, c.k, c.k, 1, c.k // 2)
        _, y = H.conv_gemm(x, c.W, geom=geom, bias=c.b32, out_bf16=False, out_f32=True)
        return y, geom

    def _conv_bwd(self, c, dy16, x, geom, need_dx=True):
        self.net.wg.run(lambda: H.conv_wgrad(dy16, x, c.gw, geom=geom, dbias=c.gb), dy16, x)
        if not need_dx:
            return None
        B, SH, SW, SC, DH, DW, N, KH, KW, s, p = geom
        _, dx = H.conv_gemm(dy16, c.WT, geom=(B, DH, DW, N, SH, SW, SC, KH, KW, s, p), transposed=True, out_bf16=False, out_f32=True)
        return dx

    def _gn(self, i, u, B, HW, C):
        P = self.store.P
        return H.gn_nhwc_fwd(u, P[f"mask_head.gn{i}.weight"], P[f"mask_head.gn{i}.bias"], B, HW, C, 8, ldy=u.shape[-1])

    def _gn_bwd(self, i, dy, u, stats, B, HW, C):
        P, G = self.store.P, self.store.G
        return H.gn_nhwc_bwd(dy, u, P[f"mask_head.gn{i}.weight"], P[f"mask_head.gn{i}.bias"], stats,
                             G[f"mask_head.gn{i}.weight"], G[f"mask_head.gn{i}.bias"], B, HW, C, 8, lddx=u.shape[-1])

    # ------------------------------------------------------------------ forward
    def forward(self, hs_last16, mem16, mem32, src32, pad_u8, feats, B, S, Lq, h, w):
        """hs_last16 [B,E] bf16 (last decoder layer, normed), mem16/mem32 [B*S,E] encoder memory, src32 [B*S,E] the
        sequence buffer that holds input_proj+GN in its image rows, pad_u8 [B,hw] (1 = padded pixel), feats = the four
        ResNet stage outputs.  Returns (pred [B,1,H1,W1] fp32 logits at stride 4, mask_att [B,nh,h,w], saved)."""
        cfg, net = self.cfg, self.net
        E, nh, HW = cfg.hidden, cfg.nheads, h * w
        cv = self.convs
        _, q = net.lin_fwd("bbox_attention.q_linear.", hs_last16, out_bf16=False, out_f32=True)
        _, kall = net.lin_fwd("bbox_attention.k_linear.", mem16, out_bf16=False, out_f32=True)
        X0 = torch.empty(B * HW, cv["lay1"].cip, dtype=torch.bfloat16, device=mem16.device)
        H.seg_concat(src32, mem32, X0, B, HW, E, nh, S, Lq)
        Pm = H.attn_map_fwd(q, kall, pad_u8, B, HW, E, nh, S, Lq, concat=X0, concat_col=2 * E)
        sv = dict(q=q, kall=kall, P=Pm, X0=X0, hs_last16=hs_last16, mem16=mem16, B=B, S=S, Lq=Lq, h=h, w=w, stages=[])
        u1, g1 = self._conv(cv["lay1"], X0, B, h, w); a1, s1 = self._gn(1, u1, B, HW, cv["lay1"].cout)
        u2, g2 = self._conv(cv["lay2"], a1, B, h, w); a2, s2 = self._gn(2, u2, B, HW, cv["lay2"].cout)
        sv.update(u1=u1, g1=g1, a1=a1, s1=s1, u2=u2, g2=g2, s2=s2)
        a, (ph, pw) = a2, (h, w)
        # FPN: stride 16, 8, 4 features (layer3, layer2, layer1 outputs)
        for j, fi in enumerate((2, 1, 0)):
            f16, (_, fh, fw) = feats[fi]
            ad, lay = cv[f"adapter{j + 1}"], cv[f"lay{j + 3}"]
            _, fo = H.linear(f16, ad.W.view(ad.cop, ad.cip), bias=ad.b32, out_bf16=False, out_f32=True)
            x = H.upsample_add(fo, a, B, fh, fw, ph, pw, ad.cout, ldo=lay.cip)
            u, g = self._conv(lay, x, B, fh, fw)
            a_new, st = self._gn(j + 3, u, B, fh * fw, lay.cout)
            sv["stages"].append(dict(f16=f16, x=x, u=u, g=g, st=st, fh=fh, fw=fw, ph=ph, pw=pw))
            a, (ph, pw) = a_new, (fh, fw)
        sv["a5"] = a
        po, gout = self._conv(cv["out_lay"], a, B, ph, pw)
        sv["gout"] = gout
        pred = po[:, 0].reshape(B, 1, ph, pw).contiguous()
        if cfg.cem:                        # CEM block on (last decoder output, res_feat) (reftr_segmentation.py:139-146)
            P = self.store.P
            sv["cem_loss"], sv["cem"] = H.cem_fwd(hs_last16, P["cem_block.c3.weight"], P["cem_block.c3.bias"], a,
                                                  P["cem_block.c2.weight"], P["cem_block.c2.bias"], B, ph * pw)
        return pred, Pm.view(B, nh, h, w), sv

    # ------------------------------------------------------------------ backward
    def backward(self, sv, dpred, dmem, dcem=None):
        """dpred [B,1,H1,W1] fp32, dcem [1] fp32 = d loss_cem (cfg.cem).  Accumulates the encoder-memory gradient into dmem ([B*S,E] fp32) and returns
        (d_hs_last fp32 [B,E], d_src fp32 [B*hw,E] (gradient of the input_proj+GN rows), extra = {stage index: ungated
        fp32 gradient w.r.t. that ResNet stage output})."""
        cfg, net = self.cfg, self.net
        E, nh = cfg.hidden, cfg.nheads
        cv = self.convs
        B, S, Lq, h, w = (sv[k] for k in ("B", "S", "Lq", "h", "w"))
        HW = h * w
        st5 = sv["stages"][-1]
        M5 = B * st5["fh"] * st5["fw"]
        d16 = torch.zeros(M5, cv["out_lay"].cop, dtype=torch.bfloat16, device=dpred.device)
        d16[:, 0] = dpred.reshape(-1)
        da = self._conv_bwd(cv["out_lay"], d16, sv["a5"], sv["gout"])
        d_hs_cem = None
        if cfg.cem:
            P, G = self.store.P, self.store.G
            d_hs_cem = H.cem_bwd(sv["hs_last16"], P["cem_block.c3.weight"], P["cem_block.c3.bias"], sv["a5"],
                                 P["cem_block.c2.weight"], P["cem_block.c2.bias"], sv["cem"], dcem, da,
                                 G["cem_block.c3.weight"], G["cem_block.c3.bias"], G["cem_block.c2.weight"], B, st5["fh"] * st5["fw"])
        extra = {}
        for j in (2, 1, 0):
            stg = sv["stages"][j]
            ad, lay = cv[f"adapter{j + 1}"], cv[f"lay{j + 3}"]
            fh, fw, ph, pw = stg["fh"], stg["fw"], stg["ph"], stg["pw"]
            du = self._gn_bwd(j + 3, da, stg["u"], stg["st"], B, fh * fw, lay.cout)
            dx = self._conv_bwd(lay, du, stg["x"], stg["g"])
            prev_ld = cv[f"lay{j + 2}"].cop
            da, dyb = H.upsample_add_bwd(dx, B, fh, fw, ph, pw, ad.cout, ldda=prev_ld, lddyb=ad.cop)
            fi = (2, 1, 0)[j]
            self.net.wg.run(lambda dyb=dyb, stg=stg, ad=ad: H.linear_wgrad(dyb, stg["f16"], ad.gw.view(ad.cop, ad.cip), dbias=ad.gb), dyb, stg["f16"])
            if fi > 0:          # layer1 is frozen: no gradient w.r.t. the stride-4 features is needed
                _, extra[fi] = H.linear(dyb, ad.WT.view(ad.cip, ad.cop), out_bf16=False, out_f32=True)
        du2 = self._gn_bwd(2, da, sv["u2"], sv["s2"], B, HW, cv["lay2"].cout)
        da1 = self._conv_bwd(cv["lay2"], du2, sv["a1"], sv["g2"])
        du1 = self._gn_bwd(1, da1, sv["u1"], sv["s1"], B, HW, cv["lay1"].cout)
        dX0 = self._conv_bwd(cv["lay1"], du1, sv["X0"], sv["g1"])
        dq, dk = H.attn_map_bwd(sv["q"], sv["kall"], sv["P"], dX0, B, HW, E, nh, S, Lq, 2 * E)
        dq16 = dq.to(torch.bfloat16); dk16 = dk.to(torch.bfloat16)
        _, d_hs = net.lin_bwd("bbox_attention.q_linear.", dq16, sv["hs_last16"], out_bf16=False, out_f32=True)
        if d_hs_cem is not None:
            d_hs += d_hs_cem
        net.lin_bwd("bbox_attention.k_linear.", dk16, sv["mem16"], res_f32=dmem, out_bf16=False, out_f32=dmem)
        d_src = dX0[:, :E].contiguous()
        d_memvis = dX0[:, E:2 * E].contiguous()
        H.rows_add(B * HW, E, a_f32=d_memvis, out_f32=dmem, accumulate=True, o_map=(HW, S, Lq))
        return d_hs, d_src, extra
