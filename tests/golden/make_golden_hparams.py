'''
Golden capture for the hyperparameter surface -- runs ONLY in the build container, where the
reference is mounted read-only at /root/reference.  Only the resulting .npz / .json (data)
travel.

What is executed from the reference: `app/hparams.py` (Hyperparameter.load_json / load / digest,
:26-69, pure Python once its top-level imports resolve), imported by file path, over the
reference's own `default.json`.  Two stand-ins, both in memory: an empty module named
`tensorflow` (imported at app/hparams.py:9, never touched by load_json / digest), and
`scipy.signal.hann = scipy.signal.windows.hann` (default.json:7 evaluates
`scipy.signal.hann(...)`, which scipy >= 1.13 only provides under `scipy.signal.windows`; same
function, symmetric by default).

TRUST: this script EXECUTES code from the untrusted reference tree -- `app/hparams.py` at import, and
`Hyperparameter.digest()` eval()s the expression strings of `default.json` (app/hparams.py:42: the
FFT window, the dtype names).  It is a build-container tool for whoever regenerates the fixture, to be
run by hand in a sandbox (no credentials, no network: this container) after reading those two files;
nothing in tests/, bench.py or the product imports it, and the GPU box never sees the reference.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_hparams.py
'''
import sys
sys.dont_write_bytecode = True      # never write __pycache__ into the reference
import importlib.util
import json
import os
import types

import numpy as np
import scipy.signal
import scipy.signal.windows

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    sys.modules.setdefault('tensorflow', types.ModuleType('tensorflow'))
    if not hasattr(scipy.signal, 'hann'):
        scipy.signal.hann = scipy.signal.windows.hann
    spec = importlib.util.spec_from_file_location('ref_hparams', os.path.join(REF, 'app', 'hparams.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    windows = {}
    out = None
    for n in (256, 512):
        hp = mod.Hyperparameter()
        hp.load_json(os.path.join(REF, 'default.json'))
        if n != 256:
            hp.load(dict(FFT_SIZE=n))
        raw = dict(hp.__dict__)
        hp.digest()
        windows['FFT_WND_%d' % n] = np.asarray(hp.FFT_WND)
        if n == 256:
            out = dict(loaded={k: v for k, v in raw.items()},
                       derived=dict(COMPLEXX=hp.COMPLEXX, FEATURE_SIZE=hp.FEATURE_SIZE,
                                    FFT_WND_dtype=str(hp.FFT_WND.dtype), FFT_WND_shape=list(hp.FFT_WND.shape)))
        else:
            out['derived_fft512'] = dict(FEATURE_SIZE=hp.FEATURE_SIZE, FFT_WND_shape=list(hp.FFT_WND.shape))
    out['registries'] = sorted(k for k in vars(mod.Hyperparameter) if k.endswith('_registry'))
    out['accessors'] = sorted(k for k in vars(mod.Hyperparameter)
                              if k.startswith(('register_', 'get_')))
    with open(os.path.join(OUT, 'hparams_ref.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    np.savez(os.path.join(OUT, 'hparams_ref.npz'), **windows)
    print(json.dumps(out, indent=1, sort_keys=True)[:600])


if __name__ == '__main__':
    main()
