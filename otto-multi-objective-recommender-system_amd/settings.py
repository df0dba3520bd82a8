"""Env-driven settings shim with the attribute names of the reference's
``src/settings.py:7-28`` (ROOT, DATA, LOGS, MODELS, EDA + a configured root logger).

The reference hard-codes ``/home/gunes/...`` and opens a log file at import
(``src/settings.py:7,19``); here the root comes from ``OTTO_ROOT`` (default: the
current working directory) and the log file is only created when ``LOGS`` exists.
"""
from datetime import datetime
from pathlib import Path
import logging
import os

ROOT = Path(os.environ.get('OTTO_ROOT', os.getcwd()))
DATA = Path(os.environ.get('OTTO_DATA', ROOT / 'data'))
LOGS = Path(os.environ.get('OTTO_LOGS', ROOT / 'logs'))
MODELS = Path(os.environ.get('OTTO_MODELS', ROOT / 'models'))
EDA = Path(os.environ.get('OTTO_EDA', ROOT / 'eda'))

LOGGING_LEVEL = logging.INFO
log_formatter = logging.Formatter(
    '%(asctime)s %(levelname)s %(module)s - %(funcName)s: %(message)s',
    datefmt='%Y-%m-%d %H:%M:%S'
)
logger = logging.getLogger('root')
logger.setLevel(LOGGING_LEVEL)
if not any(isinstance(h, logging.StreamHandler) for h in logger.handlers):
    log_stream_handler = logging.StreamHandler()
    log_stream_handler.setFormatter(log_formatter)
    log_stream_handler.setLevel(LOGGING_LEVEL)
    logger.addHandler(log_stream_handler)
if LOGS.is_dir() and not any(isinstance(h, logging.FileHandler) for h in logger.handlers):
    log_file_handler = logging.FileHandler(filename=LOGS / f'{datetime.now().strftime("%Y-%m-%d %H:%M:%S")}.log')
    log_file_handler.setFormatter(log_formatter)
    log_file_handler.setLevel(LOGGING_LEVEL)
    logger.addHandler(log_file_handler)
