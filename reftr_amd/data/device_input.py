"""The deterministic part of the reference's input pipeline on the device: RandomResize([size],
max_size) -> ToTensor -> Normalize of datasets/refer_resc.py:100-121 / datasets/transforms.py, and the batch padding of
util/collate_fn.py:24-41, fed from uint8 HWC images (pinned host memory or device).  Output is the `samples['img']`
NestedTensor and the updated targets (boxes xyxy pixels -> cxcywh normalised, 'size', nearest-resized 'masks') the model
and the criterion expect.  Stochastic augmentations (RandomIntensitySaturation, cv2 HSV jitter) stay on the host: they
run before the resize on the raw image."""
import torch

from .. import hip as H
from ..util.misc import NestedTensor
from . import resample

MEAN = (0.485, 0.456, 0.406)      # datasets/refer_resc.py:103
STD = (0.229, 0.224, 0.225)


class DeviceInputPipeline:
    def __init__(self, size, max_size=None, mean=MEAN, std=STD, device="cuda"):
        self.size, self.max_size, self.mean, self.std = size, max_size, mean, std
        self.device = torch.device(device)
        self._taps = {}
        self.copy_stream = torch.cuda.Stream(device=self.device)

    def _dev_taps(self, n_in, n_out):
        key = (n_in, n_out)
        if key not in self._taps:
            b, c = resample.taps(n_in, n_out)
            self._taps[key] = (b.to(self.device), c.to(self.device))
        return self._taps[key]

    def resize(self, img, oh, ow):
        """uint8 [H, W, 3] on the device -> uint8 [oh, ow, 3]; horizontal pass first, like Pillow."""
        Hh, Ww = img.shape[:2]
        if ow != Ww:
            img = H.resample_u8(img, *self._dev_taps(Ww, ow), ow, axis=1)
        if oh != Hh:
            img = H.resample_u8(img, *self._dev_taps(Hh, oh), oh, axis=0)
        return img

    def __call__(self, images, targets=None):
        """images: list of uint8 [H, W, 3] tensors (host, ideally pinned, or device).  Returns (NestedTensor, targets)."""
        cur = torch.cuda.current_stream(self.device)
        self.copy_stream.wait_stream(cur)
        dev_imgs = []
        with torch.cuda.stream(self.copy_stream):          # H2D of the raw bytes: 4x fewer than the fp32 tensors
            for im in images:
                assert im.dtype == torch.uint8 and im.dim() == 3 and im.shape[2] == 3
                dev_imgs.append(im.to(self.device, non_blocking=True).contiguous())
        cur.wait_stream(self.copy_stream)
        resized, out_t = [], []
        for i, im in enumerate(dev_imgs):
            im.record_stream(cur)
            Hh, Ww = im.shape[:2]
            oh, ow = resample.size_with_aspect_ratio(Ww, Hh, self.size, self.max_size)
            resized.append(self.resize(im, oh, ow))
            if targets is not None:
                out_t.append(self._target(targets[i], (Hh, Ww), (oh, ow)))
        Hm = max(r.shape[0] for r in resized); Wm = max(r.shape[1] for r in resized)
        batch, mask = H.img_collate_norm(resized, Hm, Wm, self.mean, self.std)
        return NestedTensor(batch, mask.bool()), (out_t if targets is not None else None)

    def _target(self, t, in_hw, out_hw):
        """datasets/transforms.py:118-137 (resize) + :252-262 (Normalize): O(1)-sized tensor arithmetic."""
        (Hh, Ww), (oh, ow) = in_hw, out_hw
        rw, rh = float(ow) / float(Ww), float(oh) / float(Hh)
        t = dict(t)
        if "boxes" in t:
            b = t["boxes"].to(torch.float32) * torch.as_tensor([rw, rh, rw, rh], device=t["boxes"].device)
            x0, y0, x1, y1 = b.unbind(-1)
            b = torch.stack([(x0 + x1) / 2, (y0 + y1) / 2, x1 - x0, y1 - y0], dim=-1)
            t["boxes"] = b / torch.tensor([ow, oh, ow, oh], dtype=torch.float32, device=b.device)
        if "area" in t:
            t["area"] = t["area"] * (rw * rh)
        t["size"] = torch.tensor([oh, ow])
        if "masks" in t:
            t["masks"] = torch.nn.functional.interpolate(t["masks"][:, None].float(), (oh, ow), mode="nearest")[:, 0] > 0.5
        return t
