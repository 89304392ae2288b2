"""decode-time throughput of greedy generation with the KV cache on a bench config; GPU box only.
  python scripts/bench_generate.py [--model cruller_large_1280x960] [--batch 8] [--steps 64]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='cruller_large_1280x960')
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--steps', type=int, default=64)
    a = ap.parse_args()
    from pixparse_amd.models import Cruller, get_model_config
    dev = torch.device('cuda:0')
    cfg = get_model_config(a.model)
    torch.manual_seed(0)
    model = Cruller(cfg, vocab_size=50267).to(dev)
    model._ensure_engines()
    model.refresh_shadows(full=True)
    B = a.batch
    image = torch.randn(B, model.in_chans, *model.img_size, device=dev)
    t0 = time.time(); enc = model.image_encoder(image); torch.cuda.synchronize(); t_enc = time.time() - t0
    t0 = time.time(); model.decode_begin(enc, a.steps + 8); torch.cuda.synchronize(); t_begin = time.time() - t0
    ids = torch.full((B, 1), 50266, dtype=torch.int64, device=dev)
    for _ in range(4):                                    # warm-up steps (they also fill cache positions 0..3)
        ids = model.decode_step(ids).float().argmax(-1, keepdim=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.steps):
        ids = model.decode_step(ids).float().argmax(-1, keepdim=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.steps
    _, dec, _ = model._engines
    w_bytes = sum(e.numel for n, e in model.arena.entries.items() if n.startswith('text_decoder.') and 'embed_positions' not in n) * 2
    kv_bytes = dec.L * B * dec.gen['S'] * 2 * dec.D * 2
    print(f'{a.model} B={B} S={dec.gen["S"]}: encoder {t_enc*1e3:.1f} ms, decode_begin {t_begin*1e3:.1f} ms, '
          f'{ms:.3f} ms/step = {B / ms * 1e3:.0f} tokens/s; per step reads >= {w_bytes/1e6:.0f} MB weights + {kv_bytes/1e6:.0f} MB cross K/V '
          f'-> {(w_bytes + kv_bytes) / ms / 1e9:.2f} TB/s effective')


if __name__ == '__main__':
    main()
