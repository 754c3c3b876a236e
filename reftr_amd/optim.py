"""FusedAdamW: torch.optim.AdamW + clip_grad_norm_ of the reference loop (main_vg.py:234-268,
engine_vg.py:62-66) as two kernels over the model's flat parameter / gradient buffers.

It is a torch.optim.Optimizer so `lr_scheduler.step()` (StepLR / LambdaLR, main_vg.py:269-287) and
`optimizer.param_groups[0]["lr"]` (engine_vg.py:71) keep working; the three param groups are the
reference's (default lr / lr_backbone names 'img_backbone.0' / lr_bert names 'lang_backbone', the latter
also at args.lr_backbone — main_vg.py:251-255).
"""
import torch

from . import hip as H
from .models import layout as L


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-4, lr_backbone=1e-5, lr_bert=None, weight_decay=1e-4, betas=(0.9, 0.999), eps=1e-8):
        self.model = getattr(model, "module", model)
        st = self.model.store
        named = dict(self.model.named_parameters())
        groups = []
        for grp, glr in ((L.GROUP_MAIN, lr), (L.GROUP_BACKBONE, lr_backbone),
                         (L.GROUP_BERT, lr_backbone if lr_bert is None else lr_bert)):
            ps = [named[n] for n, _, k in st.table if k == "param" and L.lr_group(n) == grp]
            groups.append({"params": ps, "lr": glr, "group_id": grp})
        super().__init__(groups, dict(lr=lr, weight_decay=weight_decay, betas=betas, eps=eps))
        dev = st.device
        self.m = torch.zeros_like(st.flat_p)
        self.v = torch.zeros_like(st.flat_p)
        self.sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.step_count = 0
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)    # device-resident step counter (graph replay)
        self._max_norm = 0.0
        self._have_sq = False

    def zero_grad(self, set_to_none=False):
        """One memset of the flat gradient buffer (param.grad views are kept)."""
        self.model.store.flat_g.zero_()

    def clip_grad_norm_(self, max_norm):
        """Launches the global-norm reduction; the clip coefficient itself is applied inside the AdamW kernel.
        Returns the device scalar that holds the total norm after step()."""
        H.sqnorm(self.model.store.flat_g, self.sq)
        self._max_norm, self._have_sq = float(max_norm), True
        return self.grad_norm

    @torch.no_grad()
    def step(self, closure=None):
        st = self.model.store
        self.step_count += 1
        H.counter_add(self.step_dev, 1)
        if not self._have_sq:
            H.sqnorm(st.flat_g, self.sq)
        ranges = []
        for g in self.param_groups:
            b, e = st.group_range[g["group_id"]]
            if e > b:
                ranges.append((b, e, g["lr"], g["weight_decay"]))
        b1, b2 = self.defaults["betas"]
        H.adamw_flat(st.flat_p, st.flat_g, self.m, self.v, step=self.step_count, ranges=ranges, gnorm_sq=self.sq,
                     gnorm_out=self.grad_norm, grad_scale=getattr(self.model, "_grad_scale", 1.0),
                     max_norm=self._max_norm, beta1=b1, beta2=b2, eps=self.defaults["eps"], step_dev=self.step_dev)
        self._have_sq = False
        self.model.mark_dirty()

    def state_dict(self):
        return {"m": self.m, "v": self.v, "step": self.step_count,
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        self.m.copy_(sd["m"]); self.v.copy_(sd["v"]); self.step_count = int(sd["step"])
        self.step_dev.fill_(self.step_count)
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)


def build_optimizer(model, args):
    """The optimizer main_vg.py:234-268 builds, on the fused kernels."""
    return FusedAdamW(model, lr=args.lr, lr_backbone=args.lr_backbone, weight_decay=args.weight_decay)
