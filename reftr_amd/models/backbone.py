"""ResNet-50/101 v1.5 body with FrozenBatchNorm folded into the convolutions — forward and hand-written
backward over the HIP implicit-GEMM kernels (NHWC bf16 activations).

Mirrors models/modeling/backbone.py:83-125 of the reference (torchvision resnet + FrozenBatchNorm2d
:43-80, conv1/layer1 frozen :87-89, pad-mask interpolation :107).  Every conv+BN(+ReLU)(+residual) is ONE
rt_conv_gemm launch: BN scale is folded into the bf16 weight, BN shift is the epilogue bias.

Backward convention: every gradient tensor that flows between blocks is dL/d(pre-ReLU value) — the ReLU
mask of a tensor is applied by the kernel that PRODUCES its gradient (`gate=` epilogue), so no separate
elementwise backward kernels run.
"""
import torch

from .. import hip as H
from . import layout as L


class ConvSpec:
    __slots__ = ("name", "bn", "cin", "cout", "k", "stride", "pad", "trainable", "dil")

    def __init__(self, name, bn, cin, cout, k, stride, trainable, dil=1):
        self.name, self.bn, self.cin, self.cout, self.k, self.stride = name, bn, cin, cout, k, stride
        self.dil = dil
        self.pad = (k // 2) * dil
        self.trainable = trainable


class BlockSpec:
    __slots__ = ("conv1", "conv2", "conv3", "down", "trainable", "first_trainable", "stage_in")


class ResNetBody:
    PFX = "img_backbone.0.body."

    def __init__(self, store, cfg):
        self.store = store
        self.cfg = cfg
        self.blocks = []          # list of stages, each a list of BlockSpec
        inpl = 64
        # --dilation (torchvision replace_stride_with_dilation=[False, False, True], backbone.py:117-125): layer4's stride becomes a
        # dilation -- its first block runs at stride 1 with the PREVIOUS dilation (1), the following blocks dilate their 3x3 by 2
        dilate = bool(getattr(cfg, "dilation", False))
        for li, n in enumerate(cfg.resnet_layers):
            planes = 64 * 2 ** li
            tr = li > 0 and getattr(cfg, "train_backbone", True)     # lr_backbone 0: nothing trains, nothing is saved for backward
            stage = []
            for bi in range(n):
                p = f"{self.PFX}layer{li + 1}.{bi}."
                s = 2 if (bi == 0 and li > 0) else 1
                dl = 1
                if dilate and li == 3:
                    s, dl = 1, (1 if bi == 0 else 2)
                b = BlockSpec()
                b.conv1 = ConvSpec(p + "conv1.weight", p + "bn1.", inpl, planes, 1, 1, tr)
                b.conv2 = ConvSpec(p + "conv2.weight", p + "bn2.", planes, planes, 3, s, tr, dil=dl)
                b.conv3 = ConvSpec(p + "conv3.weight", p + "bn3.", planes, planes * 4, 1, 1, tr)
                b.down = ConvSpec(p + "downsample.0.weight", p + "downsample.1.", inpl, planes * 4, 1, s, tr) if bi == 0 else None
                b.trainable = tr
                b.first_trainable = (li == 1 and bi == 0)     # its input (layer1 output) needs no gradient
                b.stage_in = li - 1 if bi == 0 and li > 0 else None     # index of the stage output that is this block's input
                stage.append(b)
                inpl = planes * 4
            self.blocks.append(stage)
        self.wg = H.SideStream(False)     # conv weight gradients stay inline (they fill the chip on their own)
        on_gpu = str(store.device).startswith("cuda")
        self.fuse_frozen = on_gpu         # frozen layer1 bottlenecks as one launch each (rt_bottleneck_fwd)
        self.fuse_stem = True             # frozen stem + max-pool as one launch (rt_stem_pool)
        self.batch = H.WgradBatch(workspace_mb=1024) if on_gpu else None      # conv weight gradients: grouped launches per stage
        self.wgs = H.SideStream(False)
        self.W = {}      # bf16 operands: name -> [N][T][C]; name + '.t' -> [C][T][N]
        self.bn = {}     # bn prefix -> (scale, shift) fp32
        self.all_convs = [c for st in self.blocks for b in st for c in (b.conv1, b.conv2, b.conv3, b.down) if c is not None]
        for c in self.all_convs:
            if c.trainable:
                store.register_overwritable(store.phys(c.name, grad=True))

    # ------------------------------------------------------------------ operands
    def refresh(self, full):
        """Rebuild bf16 operands from the fp32 masters.  full=True also re-folds the FrozenBN buffers and the
        frozen weights (needed after load_state_dict / device moves); the per-step call only touches the
        trainable convolutions."""
        st, P, dev = self.store, self.store.P, self.store.device
        import torch
        if full:
            bns = [self.PFX + "bn1."] + [c.bn for c in self.all_convs]
            for bn in bns:
                c = P[bn + "weight"].numel()
                if bn in self.bn and self.bn[bn][0].device == P[bn + "weight"].device:
                    sc, sh = self.bn[bn]            # refilled in place: the optimizer's operand tables hold these pointers
                else:
                    sc = torch.empty(c, dtype=torch.float32, device=dev); sh = torch.empty(c, dtype=torch.float32, device=dev)
                H.bn_fold(P[bn + "weight"], P[bn + "bias"], P[bn + "running_mean"], P[bn + "running_var"], 1e-5, sc, sh)
                self.bn[bn] = (sc, sh)
            self.W["stem"] = torch.empty(64, 7, 8, 4, dtype=torch.bfloat16, device=dev)
            H.stem_weight_prep(st.phys(self.PFX + "conv1.weight"), self.bn[self.PFX + "bn1."][0], self.W["stem"])
        for c in self.all_convs:
            T = c.k * c.k
            if c.name not in self.W:
                self.W[c.name] = torch.empty(c.cout, T, c.cin, dtype=torch.bfloat16, device=dev)
                if c.trainable:
                    self.W[c.name + ".t"] = torch.empty(c.cin, T, c.cout, dtype=torch.bfloat16, device=dev)
        if full:       # the BN scale tensors were re-created: rebuild both job tables
            self._prep_all = H.WeightPrepBatch(dev)
            self._prep_train = H.WeightPrepBatch(dev)
            for c in self.all_convs:
                args = (st.phys(c.name), c.cout, c.k * c.k, c.cin)
                kw = dict(scale=self.bn[c.bn][0], dst=self.W[c.name], dst_t=self.W.get(c.name + ".t"))
                self._prep_all.add(*args, **kw)
                if c.trainable:
                    self._prep_train.add(*args, **kw)
            self._prep_all.run()
        else:
            self._prep_train.run()

    # ------------------------------------------------------------------ forward
    def _conv(self, x, shp, c, relu, res=None):
        B, Hh, Ww = shp
        Ho = (Hh + 2 * c.pad - c.dil * (c.k - 1) - 1) // c.stride + 1
        Wo = (Ww + 2 * c.pad - c.dil * (c.k - 1) - 1) // c.stride + 1
        geom = (B, Hh, Ww, c.cin, Ho, Wo, c.cout, c.k, c.k, c.stride, c.pad)
        y, _ = H.conv_gemm(x, self.W[c.name], geom=geom, bias=self.bn[c.bn][1], res_bf16=res, res_first=True,
                           act=H.ACT_RELU if relu else H.ACT_NONE, dil=c.dil)
        return y, (B, Ho, Wo), geom

    def forward(self, img, ready=None, after_stem=None):
        """img fp32 [B,3,H,W] -> (list of the 4 stage outputs as ([M, C] bf16, (B,h,w))), saved-for-backward).
        `ready`: event after which the TRAINABLE convolutions' operands are current (the frozen stem / layer1 do not wait)."""
        B, _, Hh, Ww = img.shape
        Ho, Wo, _, _ = H.stem_geometry(Hh, Ww)
        xp = H.img_pack(img)
        if self.fuse_stem:                          # conv1 / bn1 / relu / maxpool are frozen: nothing of them is saved for backward
            y = H.stem_pool(xp, self.W["stem"], self.bn[self.PFX + "bn1."][1], Ho, Wo)
        else:
            y = H.stem_conv(xp, self.W["stem"], self.bn[self.PFX + "bn1."][1], Ho, Wo)
            y = H.maxpool3x3s2(y)
        H.mark("ResNet forward: stem + max-pool done")
        if after_stem is not None:
            after_stem()
        shp = (B, y.shape[1], y.shape[2])
        x = y.view(-1, 64)
        feats, saved = [], []
        for stage in self.blocks:
            for b in stage:
                if ready is not None and b.trainable:
                    torch.cuda.current_stream().wait_event(ready)
                    ready = None
                if self.fuse_frozen and not b.trainable and b.conv1.cout == 64 and b.conv2.stride == 1 and (b.down is None or b.conv1.cin == 64):
                    # frozen layer1 block: one launch, h1 / h2 never reach HBM (nothing of it is read by a backward)
                    Bn, Hh, Ww = shp
                    c1, c2, c3, cd = b.conv1, b.conv2, b.conv3, b.down
                    out = H.bottleneck_fwd(x.view(Bn, Hh, Ww, c1.cin), self.W[c1.name], self.bn[c1.bn][1], self.W[c2.name], self.bn[c2.bn][1],
                                           self.W[c3.name], self.bn[c3.bn][1],
                                           wd=self.W[cd.name] if cd is not None else None, bd=self.bn[cd.bn][1] if cd is not None else None)
                    x = out.view(-1, c3.cout)
                    continue
                rec = {"x": x, "shp": shp}
                idt = x
                if b.down is not None:
                    idt, _, rec["gd"] = self._conv(x, shp, b.down, relu=False)
                h1, s1, rec["g1"] = self._conv(x, shp, b.conv1, relu=True)
                h2, s2, rec["g2"] = self._conv(h1, s1, b.conv2, relu=True)
                out, s3, rec["g3"] = self._conv(h2, s2, b.conv3, relu=True, res=idt)
                if b.trainable:
                    rec.update(h1=h1, h2=h2, out=out)
                    saved.append((b, rec))
                x, shp = out, s3
            feats.append((x, shp))
            H.mark(f"ResNet forward: layer{len(feats)} done")
        return feats, saved

    # ------------------------------------------------------------------ backward
    def _wgrad(self, g, x, c, geom):
        dw, sc = self.store.phys(c.name, grad=True), self.bn[c.bn][0]
        ow = self.store.claim(dw)
        if self.batch is not None:
            self.batch.add_conv(g, x, dw, geom, scale=sc, overwrite=ow, dil=c.dil)
        else:
            self.wg.run(lambda: H.conv_wgrad(g, x, dw, geom=geom, scale=sc, overwrite=ow, dil=c.dil), g, x)

    def _dgrad(self, g, c, geom, res=None, gate=None, res_f32=None):
        B, SH, SW, SC, DH, DW, N, KH, KW, s, p = geom
        geom_t = (B, DH, DW, N, SH, SW, SC, KH, KW, s, p)
        y, _ = H.conv_gemm(g, self.W[c.name + ".t"], geom=geom_t, transposed=True, res_bf16=res, res_f32=res_f32, gate=gate, dil=c.dil)
        return y

    def backward(self, saved, g_out, extra=None):
        """g_out: bf16 [M, 2048] = dL/d(pre-ReLU of the layer4 output) (already gated by the producer).
        extra: {stage index: fp32 [M, C]} additional UNGATED gradients w.r.t. intermediate stage outputs (the RES head's
        FPN adapters read layer2 / layer3 outputs); they join the residual sum before the ReLU gate."""
        for _ in self.backward_stages(saved, g_out, extra):
            pass

    def backward_stages(self, saved, g_out, extra=None):
        """backward() as a generator: yields the stage number (4, 3, 2) when that stage's data AND weight gradients have been
        launched (its slice of the gradient buffer is final once the stream reaches this point)."""
        extra = extra or {}
        try:
            yield from self._backward(saved, g_out, extra)
        finally:
            self.wgs.join()

    def _backward(self, saved, g_out, extra):
        for b, rec in reversed(saved):
            x, h1, h2 = rec["x"], rec["h1"], rec["h2"]
            self._wgrad(g_out, h2, b.conv3, rec["g3"])
            g_h2 = self._dgrad(g_out, b.conv3, rec["g3"], gate=h2)
            self._wgrad(g_h2, h1, b.conv2, rec["g2"])
            g_h1 = self._dgrad(g_h2, b.conv2, rec["g2"], gate=h1)
            self._wgrad(g_h1, x, b.conv1, rec["g1"])
            if b.down is not None:
                self._wgrad(g_out, x, b.down, rec["gd"])
            if self.batch is not None and b.down is not None:
                # end of a stage: its queued weight gradients go out together -- on the side stream when enabled, under the
                # next stage's backward-data chain
                alive = [t for k in self.batch.keep for t in k if t is not None]      # until the side stream is joined
                self.wgs.run(self.batch.run, *alive)
            if b.down is not None:
                yield b.stage_in + 2                    # layer number (4, 3, 2) whose gradients are all launched now
            if b.first_trainable:
                break                                   # layer1 is frozen: no gradient w.r.t. its output
            g_idt = self._dgrad(g_out, b.down, rec["gd"]) if b.down is not None else g_out
            g_out = self._dgrad(g_h1, b.conv1, rec["g1"], res=g_idt, gate=x, res_f32=extra.get(b.stage_in))
