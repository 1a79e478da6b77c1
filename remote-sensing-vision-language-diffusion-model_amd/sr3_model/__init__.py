"""Stage 1 of the pipeline: ``create_model(opt)`` -> DDPM (reference: models/sr3_model/__init__.py:5-9)."""
import logging


def create_model(opt):
    """Factory with the reference's name and log line; the class is imported lazily so that importing the package
    does not load the HIP library."""
    from . import model as _model
    net = _model.DDPM(opt)
    logging.getLogger("base").info("Model [%s] is created.", type(net).__name__)
    return net
