'''
danet-tensorflow_amd: MI355X-native (gfx950) Deep Attractor Network hot path
behind the Encoder / Estimator / Separator plugin API and hparams surface of
khaotik/DaNet-Tensorflow (app/modules.py + app/ops.py).

The directory name contains a hyphen, so import it with
    importlib.import_module('danet-tensorflow_amd')
(`__graft_entry__.load_package()` does this and aliases it as `danet_amd`).
'''
from .hparams import hparams, Hyperparameter  # noqa: F401

__all__ = ['hparams', 'Hyperparameter']
