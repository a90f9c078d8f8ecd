"""Name -> class registry, like the reference's model/__init__.py:24-38."""
import logging

from .resunet import ResUNet2, ResUNetBN2C

MODELS = [ResUNetBN2C]


def load_model(name):
    mdict = {m.__name__: m for m in MODELS}
    if name not in mdict:
        logging.info(f'Invalid model index. You put {name}. Options are: {sorted(mdict)}')
        return None
    return mdict[name]
