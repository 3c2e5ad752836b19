'''
Golden-vector capture -- runs ONLY in the build container, where the reference
is mounted read-only at /root/reference.  Nothing here travels to the GPU box
except the resulting .npz fixtures (data only: inputs + reference outputs).

What is executed from the reference: `app/utils.py` (istft, random_zeropad, load_wavfile,
save_wavfile) and
its STFT call expression `scipy.signal.stft(x, window=FFT_WND, nperseg=N,
noverlap=N-S)[2].astype(COMPLEXX).T` (app/utils.py:117-122; same call at
app/datasets/TIMIT/process.py:93-97).  The model path needs TensorFlow 1.x and
cannot run -- no model-path fixture comes from the reference.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
'''
import sys
sys.dont_write_bytecode = True      # never write __pycache__ into the reference
import os
import types
import random
import importlib.util

import numpy as np
import scipy.signal
import scipy.signal.windows

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference_utils(fft_size, fft_stride, smprate=8000):
    '''import /root/reference/app/utils.py by path with in-memory parents'''
    hp = types.SimpleNamespace()
    hp.FFT_SIZE = fft_size
    hp.FFT_STRIDE = fft_stride
    hp.SMPRATE = smprate
    hp.FLOATX = 'float32'
    hp.COMPLEXX = 'complex64'
    # default.json:7 with scipy.signal.hann -> scipy.signal.windows.hann
    hp.FFT_WND = np.sqrt(scipy.signal.windows.hann(fft_size)).astype('float32')
    app = types.ModuleType('app')
    app.__path__ = []
    app_hp = types.ModuleType('app.hparams')
    app_hp.hparams = hp
    sys.modules['app'] = app
    sys.modules['app.hparams'] = app_hp
    spec = importlib.util.spec_from_file_location(
        'app.utils', os.path.join(REF, 'app', 'utils.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, hp


def ref_stft(x, hp):
    # the reference's call expression, argument for argument
    Zxx = scipy.signal.stft(
        x, window=hp.FFT_WND, nperseg=hp.FFT_SIZE,
        noverlap=hp.FFT_SIZE - hp.FFT_STRIDE)[2]
    return Zxx.astype(hp.COMPLEXX).T


def main():
    out = {}
    # ---- N=256 / S=64 -----------------------------------------------------
    utils, hp = load_reference_utils(256, 64)
    out['wnd256_bits'] = hp.FFT_WND.view(np.uint32)                 # G1
    for L in (256, 257, 319, 320, 8000, 8001, 8063, 8064):          # G2
        x = np.random.RandomState(0).randn(L).astype(np.float32)
        X = ref_stft(x, hp)
        out['stft256_L%d' % L] = X
        out['istft256_L%d' % L] = utils.istft(                      # G3
            X, stride=hp.FFT_STRIDE, window=hp.FFT_WND)
    # int16-scale input like TIMIT/process.py:44-46 feeds
    x = (np.random.RandomState(1).randn(4000) * 1000).astype(np.float32)
    out['stft256_int16scale_x'] = x
    out['stft256_int16scale'] = ref_stft(x, hp)
    # Ls < N must be an error (K10)
    try:
        ref_stft(np.zeros(255, np.float32), hp)
        out['stft256_short_raises'] = np.array(0)
    except ValueError:
        out['stft256_short_raises'] = np.array(1)
    # G4: random_zeropad under random.seed(k)
    base = np.arange(12, dtype=np.float32).reshape(3, 4)
    for k in range(4):
        random.seed(k)
        out['zeropad_seed%d_axis0' % k] = utils.random_zeropad(base, 5, axis=0)
        random.seed(k)
        out['zeropad_seed%d_axis-1' % k] = utils.random_zeropad(base, 7, axis=-1)
    out['zeropad_base'] = base

    # ---- N=512 / S=128, 10 s @16 kHz (cfg 5) --------------------------------
    utils5, hp5 = load_reference_utils(512, 128, 16000)
    out['wnd512_bits'] = hp5.FFT_WND.view(np.uint32)
    x = np.random.RandomState(0).randn(160000).astype(np.float32)
    X = ref_stft(x, hp5)
    out['stft512_L160000_shape'] = np.asarray(X.shape)
    out['stft512_L160000_head'] = X[:2]
    out['stft512_L160000_tail'] = X[-2:]
    out['stft512_L160000_mid'] = X[600:602]
    out['stft512_L160000_abs_sum'] = np.asarray(np.abs(X.astype(np.complex128)).sum())
    out['stft512_L160000_sum'] = np.asarray(X.astype(np.complex128).sum())
    y = utils5.istft(X, stride=hp5.FFT_STRIDE, window=hp5.FFT_WND)
    out['istft512_L160000_len'] = np.asarray(len(y))
    out['istft512_L160000_head'] = y[:1024]
    out['istft512_L160000_tail'] = y[-1024:]
    out['istft512_L160000_sum'] = np.asarray(y.sum())
    out['istft512_L160000_abs_sum'] = np.asarray(np.abs(y).sum())

    # ---- G6 / G7: the demo path's wav I/O (app/utils.py:95-135) -------------------
    # load_wavfile: int16 wav at the model's rate and at two others (FFT resampling to 8 kHz
    # with scipy.signal.resample, ceil'ed length) -> STFT; save_wavfile: iSTFT -> wav
    import tempfile
    import scipy.io.wavfile
    utils, hp = load_reference_utils(256, 64)
    tmp = tempfile.mkdtemp(prefix='danet_golden_')
    for rate, n in ((8000, 3000), (11025, 4100), (16000, 6001)):
        rs = np.random.RandomState(rate)
        t = np.arange(n) / float(rate)
        wave = (3000.0 * np.sin(2 * np.pi * 440.0 * t) * (0.5 + 0.5 * np.sin(2 * np.pi * 3.0 * t))
                + 300.0 * rs.randn(n)).astype(np.int16)
        fn = os.path.join(tmp, 'in_%d.wav' % rate)
        scipy.io.wavfile.write(fn, rate, wave)
        out['wav_in_%d' % rate] = wave
        out['wav_load_%d' % rate] = utils.load_wavfile(fn)
    feat = out['wav_load_11025']
    fn = os.path.join(tmp, 'out.wav')
    utils.save_wavfile(fn, feat)
    sr, back = scipy.io.wavfile.read(fn)
    out['wav_save_rate'] = np.asarray(sr)
    out['wav_save_dtype'] = np.asarray(str(back.dtype))
    out['wav_save_data'] = back

    path = os.path.join(OUT, 'frontend_ref.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes;', len(out), 'arrays')
    for m in ('app', 'app.hparams', 'app.utils'):
        sys.modules.pop(m, None)


if __name__ == '__main__':
    main()
