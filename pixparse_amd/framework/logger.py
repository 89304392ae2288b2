import logging


def setup_logging(log_file=None, level=logging.INFO):
    """ref: framework/logger.py:4-32"""
    formatter = logging.Formatter('%(asctime)s | %(levelname)s | %(message)s', datefmt='%Y-%m-%d,%H:%M:%S')
    logging.root.setLevel(level)
    logging.root.handlers = []
    sh = logging.StreamHandler()
    sh.setFormatter(formatter)
    logging.root.addHandler(sh)
    if log_file:
        fh = logging.FileHandler(filename=log_file)
        fh.setFormatter(formatter)
        logging.root.addHandler(fh)
