"""f-1 throughput: uint8 A4 page scans -> normalised fp32 [3, 1280, 960] on the GPU (crl_image_preprocess_u8), pages/s and
effective HBM bandwidth; GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixparse_amd.data import GpuImagePreprocess

dev = torch.device('cuda:0')
pre = GpuImagePreprocess((1280, 960), (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711), 3, dev)
for (H, W) in ((1754, 1240), (3508, 2480)):      # A4 at 150 / 300 dpi
    img = torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device=dev)
    out = torch.empty(3, 1280, 960, device=dev)
    for _ in range(3):
        pre(img, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        pre(img, out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    traffic = H * W * 3 + 2 * 3 * H * 960 * 4 + 3 * 1280 * 960 * 4       # u8 in, fp32 intermediate written + read, fp32 out
    print(f'{H}x{W}x3 -> 3x1280x960: {ms * 1e3:.1f} us/page = {1e3 / ms:.0f} pages/s, {traffic / 1e6:.1f} MB algorithmic -> {traffic / ms / 1e9:.2f} TB/s')
