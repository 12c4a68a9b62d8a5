"""Drop-in for the reference's ``model/LFT.py`` plugin: ``--model_name LFT`` makes the reference's
train.py:31-33 / test.py:29-31 do ``importlib.import_module('model.LFT')`` and call
``get_model(args)``, ``get_loss(args)``, ``weights_init``.  The implementation lives in lft_amd
(HIP kernels behind liblft_hip.so)."""
from lft_amd.module import get_loss, get_model, weights_init  # noqa: F401
