"""GPU image preprocessing for the train loader: decoded uint8 page -> normalised fp32 CHW at the
model's image size, same arithmetic as the reference's torchvision Compose (task_cruller_pretrain.py:132-143).
Host side = the aten `upsample_bicubic2d_aa` filter tables (pure index / weight arithmetic, cached per size pair);
device side = crl_image_preprocess_u8."""
import math
from functools import lru_cache

import torch

from .. import hip


def _cubic(x: float, a: float = -0.5) -> float:
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0
    if x < 2.0:
        return (((x - 5.0) * x + 8.0) * x - 4.0) * a
    return 0.0


@lru_cache(maxsize=64)
def aa_bicubic_tables(in_size: int, out_size: int):
    """(xmin int32[out], xsize int32[out], weights f32[out, kmax]) exactly as aten computes them for float tensors."""
    scale = in_size / out_size
    support = 2.0 * scale if scale >= 1.0 else 2.0
    invscale = 1.0 / scale if scale >= 1.0 else 1.0
    kmax = int(math.ceil(support)) * 2 + 1
    xmin = torch.zeros(out_size, dtype=torch.int32)
    xsize = torch.zeros(out_size, dtype=torch.int32)
    w = torch.zeros(out_size, kmax, dtype=torch.float32)
    f32 = lambda v: float(torch.tensor(v, dtype=torch.float32))
    scale32, support32, inv32 = f32(scale), f32(support), f32(invscale)
    for i in range(out_size):
        center = f32(scale32 * (i + 0.5))
        lo = max(int(center - support32 + 0.5), 0)
        n = min(int(center + support32 + 0.5), in_size) - lo
        ws = [_cubic(f32((j + lo - center + 0.5) * inv32)) for j in range(n)]
        tot = sum(ws)
        xmin[i], xsize[i] = lo, n
        w[i, :n] = torch.tensor([v / tot for v in ws], dtype=torch.float32)
    return xmin, xsize, w


class GpuImagePreprocess:
    """callable(image_u8 [H, W, C] torch.uint8, host or device) -> fp32 [C, Ho, Wo] on the device"""

    def __init__(self, image_size, mean, std, num_chs, device):
        self.Ho, self.Wo = int(image_size[0]), int(image_size[1])
        self.C, self.device = num_chs, device
        m = list(mean) if isinstance(mean, (tuple, list)) else [mean] * num_chs
        s = list(std) if isinstance(std, (tuple, list)) else [std] * num_chs
        self.mean = torch.tensor(m, dtype=torch.float32, device=device)
        self.std = torch.tensor(s, dtype=torch.float32, device=device)
        self._tab = {}

    def _tables(self, H, W):
        key = (H, W)
        if key not in self._tab:
            xt = [t.to(self.device) for t in aa_bicubic_tables(W, self.Wo)]
            yt = [t.to(self.device) for t in aa_bicubic_tables(H, self.Ho)]
            self._tab[key] = (xt, yt)
        return self._tab[key]

    def __call__(self, img: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        assert img.dtype == torch.uint8 and img.dim() == 3 and img.shape[2] == self.C, 'expected a uint8 [H, W, C] page'
        img = img.to(self.device, non_blocking=True).contiguous()
        H, W, C = img.shape
        (xmin, xsize, xw), (ymin, ysize, yw) = self._tables(H, W)
        tmp = torch.empty(C, H, self.Wo, dtype=torch.float32, device=self.device)
        if out is None:
            out = torch.empty(C, self.Ho, self.Wo, dtype=torch.float32, device=self.device)
        hip.call('crl_image_preprocess_u8', img.data_ptr(), H, W, C, xmin.data_ptr(), xsize.data_ptr(), xw.data_ptr(), xw.shape[1],
                 ymin.data_ptr(), ysize.data_ptr(), yw.data_ptr(), yw.shape[1], self.mean.data_ptr(), self.std.data_ptr(),
                 tmp.data_ptr(), out.data_ptr(), self.Ho, self.Wo, torch.cuda.current_stream().cuda_stream)
        return out
