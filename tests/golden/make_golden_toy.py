'''
Golden-vector capture for the `toy` dataset generator -- runs ONLY in the build container,
where the reference is mounted read-only at /root/reference.  Only the resulting .npz (data:
seeds in, arrays out) travels.

What is executed from the reference: `app/datasets/dataset.py` (WhiteNoiseData.epoch,
:43-63, pure numpy), imported by file path with an in-memory stand-in for `app.hparams`
that provides exactly what the module touches: the `register_dataset` decorator,
FEATURE_SIZE and FLOATX.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_toy.py
'''
import sys
sys.dont_write_bytecode = True      # never write __pycache__ into the reference
import os
import types
import importlib.util

import numpy as np

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference_dataset(feature_size):
    hp = types.SimpleNamespace(FEATURE_SIZE=feature_size, FLOATX='float32', registry={})

    def register_dataset(name):
        def wrapper(cls):
            hp.registry[name] = cls
            return cls
        return wrapper
    hp.register_dataset = register_dataset
    app = types.ModuleType('app')
    app.__path__ = []
    app_hp = types.ModuleType('app.hparams')
    app_hp.hparams = hp
    sys.modules['app'] = app
    sys.modules['app.hparams'] = app_hp
    spec = importlib.util.spec_from_file_location(
        'app.datasets.dataset', os.path.join(REF, 'app', 'datasets', 'dataset.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, hp


def main():
    out = {}
    # small feature size: every batch of one epoch, bit for bit
    mod, hp = load_reference_dataset(5)
    assert list(hp.registry) == ['toy']
    ds = hp.registry['toy']()
    try:
        next(ds.epoch('train', 4))
        out['unloaded_raises'] = np.array(0)
    except RuntimeError:
        out['unloaded_raises'] = np.array(1)
    ds.install_and_load()
    np.random.seed(1337)                      # the reference's own seed (WSJ0/process.py:18)
    ep = [b for (b,) in ds.epoch('train', 4, shuffle=True)]
    out['toy_f5_seed1337'] = np.stack(ep)                         # [10, 4, 128, 5] float32
    ep2 = [b for (b,) in ds.epoch('valid', 4)]                    # the stream simply continues
    out['toy_f5_seed1337_second_epoch_first'] = ep2[0]
    # cfg-1 shape (F = 129, batch B*C = 8): count, dtype, checksums, first and last batch rows
    mod, hp = load_reference_dataset(129)
    ds = hp.registry['toy']()
    ds.install_and_load()
    np.random.seed(1337)
    ep = [b for (b,) in ds.epoch('train', 8)]
    out['toy_f129_n_batches'] = np.array(len(ep))
    out['toy_f129_shape'] = np.array(ep[0].shape)
    out['toy_f129_dtype'] = np.array(str(ep[0].dtype))
    out['toy_f129_sums'] = np.array([b.astype(np.float64).sum() for b in ep])
    out['toy_f129_first_rows'] = ep[0][:, 0, :]
    out['toy_f129_last_rows'] = ep[-1][:, -1, :]
    path = os.path.join(OUT, 'toy_ref.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes;', len(out), 'arrays')
    for m in ('app', 'app.hparams', 'app.datasets.dataset'):
        sys.modules.pop(m, None)


if __name__ == '__main__':
    main()
