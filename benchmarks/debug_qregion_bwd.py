"""Debug aid (GPU box): the QueryEncoder backward, launched chain vs rt_qenc_bwd, several shapes in ONE process (as pytest runs
them); every queued weight-gradient job is recorded and checked against a torch product."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_model_gpu import build, rel
from test_qregion_gpu import _qenc_inputs
from reftr_amd import hip

E = 256
for (B, Lq, Pn, train, with_gb) in [(8, 40, 1, True, False), (2, 12, 1, False, False), (2, 90, 16, True, True), (3, 128, 5, True, True)]:
    model, crit, P, ocfg = build(small=True)
    model.train(train)
    net, st = model.net, model.store
    model.refresh_now()
    mem32, mem16, ctx, cat16, qmask, S = _qenc_inputs(model, B, Lq, 20, Pn, seed=1)
    hip.set_seed_dev(model.seed_dev)
    net.begin_step(train)
    o = model._qenc_fwd_fused(mem16, mem32, ctx, cat16, B, S, Lq, Pn)
    sv = dict(Nf=B * Pn, fq_ctx=o["fq_ctx"], co=o["co"], cst=(o["cmean"], o["crstd"]), c16=o["c16"], cls16=o["cls16"], lang16=o["lang16"],
              kq=o["kq"], qs=o["qs"], vs=o["vs"], qw=o["qw"])
    g = torch.Generator(device="cuda").manual_seed(5)
    N = B * Pn
    ga = torch.randn(N, E, device="cuda", generator=g) * 1e-2
    gb = torch.randn(N, E, device="cuda", generator=g) * 1e-2 if with_gb else None
    dqpos = torch.randn(N, E, device="cuda", generator=g) * 1e-2
    dmem0 = torch.randn(B * S, E, device="cuda", generator=g) * 1e-2
    jobs = []
    for batch in (net.small_wg, net.big_wg):
        orig = batch.add
        def rec(dy, x, dw, dbias=None, overwrite=False, _o=orig):
            jobs.append((dy, x, dw, dbias))
            return _o(dy, x, dw, dbias, overwrite=overwrite)
        batch.add = rec
    for mode in ("chain", "fused"):
        jobs.clear()
        st.flat_g.zero_()
        dmem = dmem0.clone()
        if mode == "chain":
            dcat, _ = model._qenc_bwd_chain(sv, ga, gb, dqpos, dmem, B, S, Lq, Pn, N)
        else:
            dcat = model._qenc_bwd_fused(sv, ga, gb, dqpos, dmem, B, S, Lq, Pn)
        net.flush_wgrads()
        torch.cuda.synchronize()
        print(f"[{B} {Lq} {Pn} {train}] {mode}: dcat finite {bool(torch.isfinite(dcat).all())} dmem finite {bool(torch.isfinite(dmem).all())}")
        for dy, x, dw, db in jobs:
            ref = dy.float().t() @ x.float()
            name = [n for n in st.G if st.G[n].data_ptr() == dw.data_ptr()]
            print(f"     {name[0] if name else '?':52s} rows {dy.shape[0]:4d} dy finite {bool(torch.isfinite(dy.float()).all())} x finite "
                  f"{bool(torch.isfinite(x.float()).all())} dw finite {bool(torch.isfinite(dw).all())} |dw - ref| {float((dw.view_as(ref) - ref).abs().max()):.3e} |ref| {float(ref.abs().max()):.3e}")
    hip.set_seed_dev(None)
