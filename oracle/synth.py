"""Seeded synthetic batches for the parity tests and the benchmark.

TEST INFRASTRUCTURE (oracle/): formula-built inputs, so fixtures and the GPU box regenerate the same
tensors without storing them.
"""
import torch

from oracle.weights import formula_tensor


def make_inputs(name, B, H, W, L, n_phrase=0, Lp=6):
    """Seeded, formula-built synthetic batch (ragged pad mask, different sentence lengths)."""
    img = formula_tensor(name + ".img", (B, 3, H, W), 1.5)
    mask = torch.zeros(B, H, W, dtype=torch.bool)
    for b in range(B):
        if b % 2 == 1:      # right / bottom padding on odd images
            hv, wv = (H * 3) // 4, (W * 2) // 3
            mask[b, hv:, :] = True; mask[b, :, wv:] = True
            img[b, :, hv:, :] = 0; img[b, :, :, wv:] = 0
    ids = torch.zeros(B, L, dtype=torch.long)
    smask = torch.zeros(B, L, dtype=torch.long)
    u = (formula_tensor(name + ".ids", (B, L), 1.0, bf16=False) * 0.5 + 0.5)
    for b in range(B):
        n = max(4, L - 3 * b - 1)
        ids[b, :n] = (1000 + (u[b, :n] * 28000)).long()
        ids[b, 0] = 101; ids[b, n - 1] = 102
        smask[b, :n] = 1
    samples = {"img": img, "img_mask": mask, "sentence": ids, "sentence_mask": smask}
    if n_phrase:
        ph = torch.zeros(B, n_phrase, Lp, dtype=torch.long)
        pm = torch.zeros(B, n_phrase, Lp, dtype=torch.long)
        pl = torch.zeros(B, n_phrase, dtype=torch.long); pr = torch.zeros(B, n_phrase, dtype=torch.long)
        for b in range(B):
            nvalid = n_phrase - b            # image b has n_phrase-b real phrases
            for j in range(n_phrase):
                if j < nvalid:
                    n = 3 + (j % 3)
                    ph[b, j, :n] = 2000 + 37 * j + torch.arange(n); ph[b, j, 0] = 101; ph[b, j, n - 1] = 102
                    pm[b, j, :n] = 1
                    pl[b, j] = 1 + j; pr[b, j] = 1 + j + (n - 2)
                else:                        # empty phrase: "[CLS] [SEP]" only -> ignored (reftr_transformer.py:235)
                    ph[b, j, 0] = 101; ph[b, j, 1] = 102; pm[b, j, :2] = 1
                    pl[b, j] = 0; pr[b, j] = 1      # grounding_datasets/refer_dataset.py:182-183
        samples.update({"phrase": ph, "phrase_mask": pm, "phrase_pos_l": pl, "phrase_pos_r": pr})
    tg = formula_tensor(name + ".boxes", (B, max(n_phrase, 1), 4), 1.0, bf16=False) * 0.5 + 0.5
    targets = []
    for b in range(B):
        n = (n_phrase - b) if n_phrase else 1
        bx = torch.stack([0.3 + 0.4 * tg[b, :n, 0], 0.3 + 0.4 * tg[b, :n, 1],
                          0.1 + 0.4 * tg[b, :n, 2], 0.1 + 0.4 * tg[b, :n, 3]], dim=-1)
        targets.append({"boxes": bx, "labels": torch.zeros(n, dtype=torch.long)})
    return samples, targets




def roberta_inputs():
    """The e2e_roberta fixture's batch: multi-phrase, RoBERTa token conventions (padding id 1, real ids >= 2)."""
    samples, targets = make_inputs("e2e_roberta", B=2, H=96, W=128, L=12, n_phrase=3)
    for k in ("sentence", "phrase"):
        m = samples[k + "_mask"].bool()
        samples[k] = torch.where(m, samples[k].clamp(min=2), torch.ones_like(samples[k]))
    return samples, targets
