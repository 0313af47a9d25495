"""Config surface of train_model.py (reference: src/scrubvae/params/param_keys.py:1-34,
params/read.py:8-42): same sections, same keys, same default-filling."""
from . import read  # noqa: F401
from .param_keys import PARAM_KEYS  # noqa: F401
