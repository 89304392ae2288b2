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
    def run(use_graph):
        torch.cuda.synchronize(); t0 = time.time()
        out = model.generate_greedy(enc, 50266, -1, a.steps, use_graph=use_graph)      # eos -1: never finishes early
        torch.cuda.synchronize()
        return (time.time() - t0) * 1e3, out
    run(True)                                              # warm-up: allocations, first-touch
    ms_e, out_e = run(False)
    ms_g, out_g = run(True)
    assert torch.equal(out_e, out_g), 'graph replay and eager steps disagree'
    print(f'eager: {ms_e / a.steps:.3f} ms/step; hipGraph replay: {ms_g / a.steps:.3f} ms/step (includes decode_begin + capture)')
    ms = ms_g / a.steps
    t_begin = 0.0
    _, dec, _ = model._engines
    w_bytes = sum(e.numel for n, e in model.arena.entries.items() if n.startswith('text_decoder.') and 'embed_positions' not in n) * 2
    kv_bytes = dec.L * B * dec.gen['S'] * 2 * dec.D * 2
    print(f'{a.model} B={B} S={dec.gen["S"]}: encoder {t_enc*1e3:.1f} ms, decode_begin {t_begin*1e3:.1f} ms, '
          f'{ms:.3f} ms/step = {B / ms * 1e3:.0f} tokens/s; per step reads >= {w_bytes/1e6:.0f} MB weights + {kv_bytes/1e6:.0f} MB cross K/V '
          f'-> {(w_bytes + kv_bytes) / ms / 1e9:.2f} TB/s effective')


if __name__ == '__main__':
    main()
