import re


def _natural_key(string_):
    """natural sort key (ref: utils/name_utils.py)"""
    return [int(s) if s.isdigit() else s for s in re.split(r'(\d+)', string_.lower())]


def clean_name(name):
    return name.replace('/', '-').replace('\\', '-')
