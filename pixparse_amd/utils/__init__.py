from .name_utils import _natural_key, clean_name
