"""pixparse_amd: the Cruller pretrain step of huggingface/pixparse, MI355X-native."""
__version__ = '0.1.0'
