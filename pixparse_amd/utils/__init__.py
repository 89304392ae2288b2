from .name_utils import _natural_key, clean_name
from .json_utils import json2token, token2json
from .ocr_utils import generate_ocr, get_cer_wer_metrics, get_generated_tokens, get_next_token, get_ocr_metrics
from .metrics import JSONParseEvaluator, average_normalized_levenshtein_similarity, edit_distance, tree_edit_distance
