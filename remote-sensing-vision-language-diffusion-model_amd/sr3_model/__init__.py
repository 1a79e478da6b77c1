"""Stage 1 of the pipeline: ``create_model(opt)`` -> DDPM (reference: models/sr3_model/__init__.py:5-9)."""
import logging

logger = logging.getLogger("base")


def create_model(opt):
    from .model import DDPM as M
    m = M(opt)
    logger.info("Model [%s] is created.", m.__class__.__name__)
    return m
