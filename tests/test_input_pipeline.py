"""CPU: the oracle of the input pipeline (oracle/input_pipeline.py) pinned against Pillow itself (the third-party
resampler behind the reference's datasets/transforms.py:resize) — bit-exact uint8 — and host-side size logic."""
import numpy as np
import pytest
import torch

from oracle import input_pipeline as IP


@pytest.mark.parametrize("H,W,oh,ow", [(37, 53, 11, 20), (40, 64, 80, 128), (61, 47, 61, 20), (33, 90, 50, 90), (128, 96, 37, 29),
                                       (17, 19, 64, 70), (200, 300, 64, 96)])
def test_resize_is_bit_exact_with_pillow(H, W, oh, ow):
    from PIL import Image
    rng = np.random.default_rng(H * 1000 + W)
    img = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
    mine = IP.pil_bilinear_resize_u8(img, oh, ow)
    assert mine.shape == ref.shape and np.array_equal(mine, ref)


def test_size_logic_known_answers():
    # datasets/transforms.py:84-104 (image_size = (w, h)); RefCOCO-style 640 / max 640 and DETR-style 800 / 1333
    assert IP.get_size_with_aspect_ratio((640, 480), 640, 640) == (480, 640)
    assert IP.get_size_with_aspect_ratio((480, 640), 640, 640) == (640, 480)
    assert IP.get_size_with_aspect_ratio((500, 375), 640, 640) == (480, 640)
    assert IP.get_size_with_aspect_ratio((500, 375), 800, 1333) == (800, 1066)
    assert IP.get_size_with_aspect_ratio((1000, 200), 800, 1333) == (267, 1335)       # round() then int(): the DETR-lineage overshoot
    assert IP.get_size_with_aspect_ratio((300, 300), 300, None) == (300, 300)


def test_normalize_boxes_and_collate():
    img = np.full((4, 6, 3), 255, dtype=np.uint8)
    x, t = IP.to_tensor_normalize(img, {"boxes": torch.tensor([[1.0, 1.0, 4.0, 3.0]])})
    assert torch.allclose(x[:, 0, 0], (1 - torch.tensor(IP.MEAN)) / torch.tensor(IP.STD))
    assert torch.allclose(t["boxes"], torch.tensor([[2.5 / 6, 2.0 / 4, 3.0 / 6, 2.0 / 4]]))
    b, m = IP.collate([torch.ones(3, 4, 6), torch.ones(3, 5, 3)])
    assert b.shape == (2, 3, 5, 6) and not m[0, :4, :6].any() and m[0, 4:].all() and m[1, :, 3:].all()


def test_product_taps_equal_oracle_taps():
    """reftr_amd/data/resample.py (host side of the product) produces the same integer taps as the oracle."""
    from reftr_amd.data import resample
    for n_in, n_out in ((53, 20), (64, 128), (300, 96), (47, 47), (480, 640), (1000, 267)):
        b, c = resample.taps(n_in, n_out)
        ref = IP.resample_coeffs(n_in, n_out)
        for o, (lo, n, k) in enumerate(ref):
            assert int(b[o, 0]) == lo and int(b[o, 1]) == n and c[o, :n].tolist() == k and not c[o, n:].any()
    for args in ((640, 480, 640, 640), (500, 375, 800, 1333), (1000, 200, 800, 1333), (300, 300, 300, None)):
        assert resample.size_with_aspect_ratio(*args) == IP.get_size_with_aspect_ratio(args[:2], args[2], args[3])


@pytest.mark.gpu
def test_device_input_pipeline_bit_exact(hip):
    """uint8 images of different sizes -> resize (Pillow-compatible) -> normalise -> pad: the resized bytes and the padding
    mask are bit-exact against the oracle (itself pinned to Pillow above); the fp32 batch to 1 ulp."""
    from reftr_amd.data import DeviceInputPipeline
    rng = np.random.default_rng(7)
    sizes = [(375, 500), (480, 640), (640, 427), (97, 211)]
    imgs = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for h, w in sizes]
    targets = [{"boxes": torch.tensor([[10.0, 20.0, w * 0.6, h * 0.7]]), "labels": torch.zeros(1, dtype=torch.long),
                "masks": torch.from_numpy(rng.integers(0, 2, size=(1, h, w)).astype(bool))} for h, w in sizes]
    pipe = DeviceInputPipeline(size=320, max_size=320)
    nt, tg = pipe([torch.from_numpy(i).pin_memory() for i in imgs], targets)
    ref_b, ref_m, ref_t = IP.preprocess_batch(imgs, targets, 320, 320)
    assert nt.tensors.shape == ref_b.shape and torch.equal(nt.mask.cpu(), ref_m)
    for i, im in enumerate(imgs):
        oh, ow = IP.get_size_with_aspect_ratio((im.shape[1], im.shape[0]), 320, 320)
        dev = pipe.resize(torch.from_numpy(im).cuda(), oh, ow)
        assert np.array_equal(dev.cpu().numpy(), IP.pil_bilinear_resize_u8(im, oh, ow)), i
        assert torch.allclose(tg[i]["boxes"], ref_t[i]["boxes"], atol=1e-6) and torch.equal(tg[i]["size"], ref_t[i]["size"])
        assert torch.equal(tg[i]["masks"], ref_t[i]["masks"])
    d = (nt.tensors.cpu() - ref_b).abs().max()
    assert float(d) < 5e-7, float(d)
