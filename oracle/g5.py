'''
CPU ORACLE fixtures "G5" -- TEST INFRASTRUCTURE ONLY.

SURVEY 8c G5: labelled, committed tensors for every `debug_fetches` key of the
reference (main.py:389-397: embed, attrs, input, output; app/modules.py:540-543:
asets, anchors, subset_choice; :571 / :600: masks) plus loss / SNR / perm idx and every
parameter gradient, at a tiny shape and at BASELINE cfg 1.

THESE ARE ORACLE-GENERATED, NOT REFERENCE-GENERATED: the reference's model path needs
TensorFlow 1.x and cannot run here, so the fixtures pin the ORACLE (a
restatement citing the reference line by line), not the reference.  Their purpose is
drift-proofing: an edit to oracle/*.py that changes any restated result fails
tests/test_oracle_cpu.py::test_g5_* until the fixtures are deliberately regenerated
(tests/golden/make_oracle_g5.py), and the HIP path is checked against the COMMITTED
arrays (tests/test_gpu_g5.py), so a same-commit edit of oracle and product cannot pass
unnoticed.

Inputs are regenerated from seeds (numpy's legacy RandomState stream is frozen across
versions); parameters are rounded to float32 (what the product holds) and evaluated in
float64.  Large tensors are stored as a strided sample plus three checksums.
'''
import numpy as np

from . import danet_oracle as O

SAMPLE_ABOVE = 20000          # elements; larger tensors are stored sampled
SAMPLE_STRIDE = 97

TINY = dict(B=2, T=8, F=5, E=3, H=4, L=2, A=4)
CASES = {
    # name: (dims, C, train_est, infer_est, separator, data)
    'tiny_truthw_sigmoid': (TINY, 2, 'truth-weighted', 'anchor', 'dot-sigmoid-orig', 'randn'),
    'tiny_anchor_softmax': (TINY, 2, 'anchor', 'anchor', 'dot-softmax-orig', 'randn'),
    'tiny_truth_softmax': (TINY, 2, 'truth', 'anchor', 'dot-softmax-orig', 'randn'),
    'tiny_truthth_sigmoid': (TINY, 2, 'truth-threshold', 'anchor', 'dot-sigmoid-orig', 'randn'),
    'tiny_anchor_softmax_c3': (TINY, 3, 'anchor', 'anchor', 'dot-softmax-orig', 'randn'),
    # BASELINE cfg 1: reference toy generator (app/datasets/dataset.py:56-58), B=4, 1x300
    'cfg1': (dict(B=4, T=128, F=129, E=20, H=300, L=1, A=6), 2, 'truth-weighted', 'anchor',
             'dot-softmax-orig', 'toy'),
}


def case_inputs(name):
    '''(src complex64 [B,C,T,F], params {tf_name: float32 ndarray}, cfg dict)'''
    dims, C, train_est, infer_est, sepn, data = CASES[name]
    B, T, F, E, H, L, A = (dims[k] for k in 'BTFEHLA')
    seed = 7000 + sorted(CASES).index(name)
    rng = np.random.RandomState(seed)
    if data == 'toy':
        src = O.toy_batch(rng, B, C, F, T=T)
    else:
        src = ((rng.randn(B, C, T, F) + 1j * rng.randn(B, C, T, F)) * 6).astype(np.complex64)
    p = O.init_bilstm_params(rng, F, E, H, L, dtype=np.float32)
    p['global/train_estimator/anchors'] = rng.randn(A, E).astype(np.float32)
    p['global/infer_estimator/anchors'] = rng.randn(A, E).astype(np.float32)
    if train_est == infer_est:
        del p['global/infer_estimator/anchors']           # main.py:256-261: one estimator
    elif train_est != 'anchor':
        del p['global/train_estimator/anchors']
    cfg = dict(H=H, L=L, E=E, C=C, A=A, train_est=train_est, infer_est=infer_est,
               separator=sepn, with_valid=True, FFT=(F - 1) * 2)
    return src, p, cfg


def case_outputs(name):
    '''every G5 tensor of the case from the LIVE oracle, float64 (ints as int64)'''
    import torch
    from . import torch_ref as R
    src, p, cfg = case_inputs(name)
    o = O.model_forward(src.astype(np.complex128), p, cfg)
    out = {k: np.asarray(o[k]) for k in
           ('embed', 'attrs', 'masks', 'output', 'sep_pwr', 'loss', 'SNR', 'perm_idx',
            'valid_attrs', 'sep_pwr_valid', 'valid_loss', 'valid_SNR', 'valid_perm_idx')}
    scope = 'train_estimator' if cfg['train_est'] == 'anchor' else 'infer_estimator'
    _, extra = O.est_anchor(o['embed'], p['global/%s/anchors' % scope].astype(np.float64),
                            cfg['C'], return_all=True)
    out['asets'] = extra['asets']                           # app/modules.py:540-543
    out['subset_choice'] = extra['subset_choice'].astype(np.int64)
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    r = R.model_forward(torch.tensor(src.astype(np.complex128)), tp, cfg)
    r['loss'].backward()
    for k in p:
        g = tp[k].grad
        out['grad:' + k] = np.zeros_like(p[k], dtype=np.float64) if g is None else g.numpy()
    return out


def _weights(n):
    return np.cos(0.37 * np.arange(n, dtype=np.float64))


def pack(arrs):
    '''{key: ndarray} -> {key or key__sample/key__sums: ndarray} (see header)'''
    out = {}
    for k, a in arrs.items():
        a = np.asarray(a)
        if a.size <= SAMPLE_ABOVE:
            out[k] = a
            continue
        flat = a.reshape(-1)
        if np.iscomplexobj(flat):
            flat = flat.view(np.float64) if flat.dtype == np.complex128 else flat.astype(np.complex128).view(np.float64)
        flat = flat.astype(np.float64)
        out[k + '__sample'] = flat[::SAMPLE_STRIDE].copy()
        out[k + '__sums'] = np.array([flat.sum(), (flat * flat).sum(), (flat * _weights(flat.size)).sum()])
        out[k + '__shape'] = np.array(a.shape, dtype=np.int64)
    return out


def compare(packed_expected, arrs, rtol):
    '''errors of live/product tensors `arrs` against a packed fixture; returns
    {key: relative error} (max |a-b| / max |b|; integer tensors: 0 if equal else inf)'''
    errs = {}
    got = pack({k: np.asarray(v, dtype=np.float64) if not np.iscomplexobj(v) and
                np.asarray(v).dtype.kind == 'f' else np.asarray(v) for k, v in arrs.items()})
    for k, b in packed_expected.items():
        if k.endswith('__shape'):
            continue
        if k not in got:
            continue
        a = got[k]
        if b.dtype.kind in 'iu':
            errs[k] = 0.0 if np.array_equal(a.astype(np.int64), b) else float('inf')
            continue
        if np.iscomplexobj(b):
            a, b = np.asarray(a).astype(np.complex128).view(np.float64), b.view(np.float64)
        if k.endswith('__sums'):
            # checksums of N terms: compare relative to the sum of magnitudes' scale
            scale = np.abs(b).max() + 1e-300
            errs[k] = float(np.abs(a - b).max() / scale)
        else:
            errs[k] = float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))
    return errs
