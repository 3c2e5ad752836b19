'''
Hyperparameter bag + plugin registries.

Mirrors the reference's `app/hparams.py:15-130` surface (same keys as
`default.json:1-41`, same `register_*` / `get_*` names, same `digest()`
derivations) so `hparams.get_encoder()` etc. resolve the MI355X-native plugins
unchanged.  Differences, all deliberate:

* `FFT_WND` is still an expression string; it is evaluated with `np`, `scipy`
  and `self` in scope like the reference (`app/hparams.py:42`), but
  `scipy.signal.hann` (removed from current scipy) is aliased to
  `scipy.signal.windows.hann`.
* Two keys are added with the reference's hard-coded values as defaults
  (`app/modules.py:212,223-242`): `NUM_LSTM_LAYERS=4`, `LSTM_HDIM=300`.
* `get_regularizer()` returns None: the reference attaches a regulariser that
  never reaches the loss (`main.py:228-229` vs `:289-290,358`).
'''
import re
import json
import types

import numpy as np
import scipy.signal
import scipy.signal.windows

DEFAULTS = {
    'FLOATX': 'float32',
    'INTX': 'int32',
    'FFT_SIZE': 256,
    'FFT_STRIDE': 64,
    'FFT_WND': 'np.sqrt(scipy.signal.hann(self.FFT_SIZE)).astype(self.FLOATX)',
    'SMPRATE': 8000,
    'BATCH_SIZE': 32,
    'MAX_N_SIGNAL': 2,
    'LENGTH_ALIGN': 4,
    'MAX_TRAIN_LEN': 128,
    'EMBED_SIZE': 20,
    'RELU_LEAKAGE': 0.3,
    'EPS': 1e-7,
    'DROPOUT_KEEP_PROB': 1.0,
    'REG_SCALE': 1e-2,
    'REG_TYPE': 'L2',
    'LR': 3e-4,
    'LR_DECAY': 0.8,
    'LR_DECAY_TYPE': None,
    'NUM_EPOCH_PER_LR_DECAY': 10,
    'GRAD_CLIP_THRES': 100.0,
    'TRAIN_ESTIMATOR_METHOD': 'truth-weighted',
    'INFER_ESTIMATOR_METHOD': 'anchor',
    'NUM_ANCHOR': 6,
    'ENCODER_TYPE': 'toy',
    'SEPARATOR_TYPE': 'dot-sigmoid-orig',
    'OPTIMIZER_TYPE': 'adam',
    'DATASET_TYPE': 'toy',
    'SUMMARY_DIR': './logs',
    'SUMMARY_TITLE': 'Test 1',
    'DEBUG': False,
    # extensions (reference values hard-coded at app/modules.py:212,223-242)
    'NUM_LSTM_LAYERS': 4,
    'LSTM_HDIM': 300,
    # k-means estimator (not in the reference, README.md:216; BASELINE cfg 5)
    'KMEANS_ITERS': 10,
}


class _ScipySignalCompat(types.SimpleNamespace):
    '''scipy.signal with the removed `hann` alias restored for FFT_WND eval.'''
    def __getattr__(self, name):
        if name == 'hann':
            return scipy.signal.windows.hann
        return getattr(scipy.signal, name)


class _ScipyCompat(types.SimpleNamespace):
    signal = _ScipySignalCompat()

    def __getattr__(self, name):
        import scipy as _sp
        return getattr(_sp, name)


class Hyperparameter:
    '''
    Contains hyperparameter settings (reference: app/hparams.py:15-127)
    '''
    pattern = r'[A-Z_]+'

    def __init__(self):
        self.__dict__.update(DEFAULTS)

    def digest(self):
        '''
        Re-derive inferred hyperparams; call after every update
        (reference: app/hparams.py:29-42).
        '''
        self.COMPLEXX = dict(
            float32='complex64', float64='complex128')[self.FLOATX]
        self.FEATURE_SIZE = 1 + self.FFT_SIZE // 2
        assert isinstance(self.DROPOUT_KEEP_PROB, float)
        assert 0. < self.DROPOUT_KEEP_PROB <= 1.
        if isinstance(self.FFT_WND, str):
            self._FFT_WND_EXPR = self.FFT_WND
        expr = getattr(self, '_FFT_WND_EXPR', None)
        if expr is not None:
            self.FFT_WND = eval(
                expr, {'np': np, 'scipy': _ScipyCompat(), 'self': self})

    def load(self, di):
        '''load from a dict (reference: app/hparams.py:44-57)'''
        assert isinstance(di, dict)
        pat = re.compile(self.pattern)
        for k, v in di.items():
            if None is pat.fullmatch(k):
                raise NameError
            assert isinstance(v, (str, int, float, bool, type(None)))
        if 'FFT_WND' in di:
            self.__dict__.pop('_FFT_WND_EXPR', None)
        self.__dict__.update(di)

    def load_json(self, file_):
        '''load from JSON file (reference: app/hparams.py:59-69)'''
        if isinstance(file_, (str, bytes)):
            file_ = open(file_, 'r')
        di = json.load(file_)
        self.load(di)

    def reset(self):
        '''back to default.json values (test helper; not in the reference)'''
        self.__dict__.clear()
        self.__dict__.update(DEFAULTS)

    def get_regularizer(self):
        # reference builds tf.contrib l1/l2 regularisers (app/hparams.py:122-127)
        # whose losses are never added to the training loss; a no-op here.
        return None


# Plugin registries (reference: app/hparams.py:72-120).  One row per plugin
# kind: (kind, registry attribute, hyperparameter that names the active
# plugin or None when the getter takes the name).  The decorators and
# getters the reference spells out one by one are generated from the table.
_PLUGIN_KINDS = (
    ('encoder', 'encoder_registry', 'ENCODER_TYPE'),
    ('estimator', 'estimator_registry', None),
    ('separator', 'separator_registry', None),
    ('optimizer', 'ozer_registry', 'OPTIMIZER_TYPE'),
    ('dataset', 'dataset_registry', 'DATASET_TYPE'),
)


def _install_plugin_kind(kind, attr, selector):
    table = {}
    setattr(Hyperparameter, attr, table)

    def register(cls, name):
        def keep(obj):
            table[name] = obj
            return obj
        return keep
    register.__name__ = 'register_' + kind
    register.__doc__ = 'decorator: add a %s plugin under `name`' % kind
    setattr(Hyperparameter, register.__name__, classmethod(register))

    if selector is None:
        def get(self, name):
            return table[name]
    else:
        def get(self):
            return table[getattr(self, selector)]
    get.__name__ = 'get_' + kind
    setattr(Hyperparameter, get.__name__, get)


for _row in _PLUGIN_KINDS:
    _install_plugin_kind(*_row)


hparams = Hyperparameter()
