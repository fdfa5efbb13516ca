"""Logger used by the mirrored modules (reference core/utils/log.py:11-18, minus the
wandb / tensorboard plumbing that is out of scope for this path)."""
import logging

LOGGER_NAME = "root"
logger = logging.getLogger(LOGGER_NAME)
if not logger.handlers:
    logger.addHandler(logging.StreamHandler())
logger.setLevel(logging.INFO)
