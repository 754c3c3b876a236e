"""Which gradients differ between the launched decoder backward and rt_decoder_bwd (and between two runs of the launched one)?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_decoder_coop_gpu as T  # noqa: E402


def diff(model, a, b, tag):
    rows = []
    st = model.store
    for n, (bk, o) in st.offset.items():
        if bk != "p":
            continue
        x, y = st.view_of(a, n).reshape(-1), st.view_of(b, n).reshape(-1)
        if not torch.equal(x, y):
            rows.append((float((x - y).abs().max()), float((x - y).norm() / (y.norm() + 1e-30)), n))
    print(f"== {tag}: {len(rows)} tensors differ")
    for d, r, n in sorted(rows, reverse=True)[:25]:
        print("   %-72s max %.2e rel %.2e" % (n, d, r))


def main():
    B, L, H, W = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (2, 2, 96, 128)))
    train = len(sys.argv) > 5 and sys.argv[5] == "train"
    model, crit, s, tg = T.build(L, B, H, W)
    a1 = T.run(model, crit, s, tg, coop=True, train=train, backward=True, coop_bwd=False)["grad"]
    a2 = T.run(model, crit, s, tg, coop=True, train=train, backward=True, coop_bwd=False)["grad"]
    b = T.run(model, crit, s, tg, coop=True, train=train, backward=True, coop_bwd=True)["grad"]
    diff(model, a1, a2, "chain vs chain")
    diff(model, a1, b, "chain vs cooperative")


if __name__ == "__main__":
    main()
