"""Published hyper-parameters of the timm / HF architectures the reference instantiates by name
(models/image_encoder_timm.py:13-20, models/text_decoder_hf.py:13) -- SURVEY App. A.1-A.3."""

VIT_ARCHS = {
    'vit_base_patch16_224': dict(patch=16, dim=768, depth=12, heads=12, mlp_ratio=4, ln_eps=1e-6, pre_norm=False,
                                 mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)),
    'vit_large_patch14_clip_224.datacompxl': dict(
        patch=14, dim=1024, depth=24, heads=16, mlp_ratio=4, ln_eps=1e-5, pre_norm=True,
        mean=(0.48145466, 0.4578275, 0.40821073), std=(0.26862954, 0.26130258, 0.27577711)),
}
SWIN_ARCHS = {
    # drop_path: timm SwinTransformer's default drop_path_rate (live in the reference: create_model leaves the encoder in train mode)
    'swin_tiny_patch4_window7_224': dict(patch=4, embed_dim=96, depths=(2, 2, 6, 2), heads=(3, 6, 12, 24), window=7,
                                         mlp_ratio=4, ln_eps=1e-5, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), drop_path=0.1),
}
BART_ARCHS = {
    'facebook/bart-base': dict(d_model=768, heads=12, ffn=3072, ln_eps=1e-5, vocab=50265, dropout=0.1, attention_dropout=0.1, activation_dropout=0.1),
    'facebook/bart-large': dict(d_model=1024, heads=16, ffn=4096, ln_eps=1e-5, vocab=50265, dropout=0.1),
}


def register_arch(kind: str, name: str, arch: dict) -> None:
    """tests and experiments register reduced geometries under their own names."""
    {'vit': VIT_ARCHS, 'swin': SWIN_ARCHS, 'bart': BART_ARCHS}[kind][name] = arch
