"""G7: the ORACLE's bf16-policy forward of BASELINE.json's headline configuration itself -- cruller_large_1280x960, all 24 encoder blocks
+ 10 decoder layers at N = 6189 / T = 1023, V = 50267, batch 1, parameters R.init_params(seed 14), sample R.synthetic_sample(seed 8) --
as a fixture: the loss, 64 sampled rows of the encoder output and six logit rows (< 1 MB).  The forward is ~8.5 TFLOP on host cores
(minutes on the GPU box's 256 cores, hours on 8): it is run ONCE here so that tests/test_realwidth_gpu.py (b') does not spend most of
the GPU suite's wall time inside the checker.  Needs only oracle/ (no reference, no GPU):

    python tests/golden/make_g7.py [out_dir]        # writes g7_cfg3_forward.safetensors + .json

The live-oracle variant of the test stays available with PIXPARSE_AMD_LIVE_ORACLE=1."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ref_cpu as R          # noqa: E402

VOCAB = 50267
VIT_L, BART_L = 'vit_large_patch14_clip_224.datacompxl', 'facebook/bart-large'
ENC_ROWS = [int(x) for x in torch.linspace(0, 6188, 64).round().tolist()]
LOGIT_ROWS = [0, 1, 17, 511, 1000, 1022]


def main(out_dir):
    from safetensors.torch import save_file
    torch.set_num_threads(os.cpu_count() or 1)
    spec = R.ModelSpec(VIT_L, BART_L, 10, 1024, (1280, 960), 3, vocab=VOCAB)
    params = R.init_params(spec, 14)
    image, tokens, target = R.synthetic_sample(spec, 1, seed=8, ragged=True)
    ti, tt = R.shift_tokens(tokens, target)
    t0 = time.time()
    with torch.no_grad():
        oenc = R.vit_forward(params, spec.enc_arch, image, 'bf16', prefix='image_encoder.trunk.', fast_attn=True)
        ologits = R.bart_decoder_forward(params, spec.dec_arch, 10, ti, oenc, 'bf16', prefix='text_decoder.trunk.', fast_attn=True)
        oloss = float(R.cross_entropy(ologits, tt))
    secs = time.time() - t0
    tensors = {'enc_rows': oenc[0, ENC_ROWS].float().contiguous(), 'logit_rows': ologits[0, LOGIT_ROWS].float().contiguous(),
               'enc_norm': oenc[0].float().norm().reshape(1), 'enc_colsum': oenc[0].float().sum(0).contiguous()}
    save_file(tensors, os.path.join(out_dir, 'g7_cfg3_forward.safetensors'))
    meta = dict(loss=oloss, enc_rows=ENC_ROWS, logit_rows=LOGIT_ROWS, param_seed=14, sample_seed=8, policy='bf16', N=int(oenc.shape[1]), T=int(ti.shape[1]),
                oracle_seconds=round(secs, 1), host_threads=torch.get_num_threads(), torch=torch.__version__,
                what='oracle/ref_cpu.py vit_forward + bart_decoder_forward + cross_entropy, cruller_large_1280x960 full depth, batch 1')
    with open(os.path.join(out_dir, 'g7_cfg3_forward.json'), 'w') as f:
        json.dump(meta, f, indent=1)
    print(json.dumps(meta))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.dirname(os.path.abspath(__file__)))
