'''
CPU ORACLE helpers -- TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py's checker leg).

Parity report of a product forward pass against the oracle at the SAME parameters,
with the float32 noise floor beside it: the reference computes in FLOATX = float32
(default.json:2), so a float32 evaluation of the reference's own arithmetic differs
from the float64 authority by rounding noise that no float32 implementation can
avoid.  The bar (north_star: "within 1e-4 relative fp32") is therefore asserted as

    err(product, f64 oracle)  <=  max(1e-4, 2 * err(f32 oracle, f64 oracle))

for every checked tensor, with err = max |a - b| / max |b| (the SURVEY-sanctioned
relative-to-tensor-max form; for masks, max |b| = 1 so it is the absolute error).
An RMS-relative figure ||a - b||_2 / ||b||_2 is reported next to it.

`logits` (the separator's pre-activation embed . attr^T, app/modules.py:587-589) is a DERIVED
entry: formed here in float64 from each path's own embedding and attractors, relative to the
largest |logit| (bar 1e-5).  It separates "error carried into the softmax" from "amplification by
the softmax": trained masks are sharp (|logit| ~ 1e3), so a logit error of 1e-6 of the maximum is an
absolute error of 1e-3 in the exponent and moves a mask by up to 2.5e-4 -- for ANY float32 path.

Reference lines restated by the functions this calls: main.py:208-337,
app/modules.py:207-260,490-603, app/ops.py:139-147,374-431 (see danet_oracle.py).
'''
import numpy as np

from . import danet_oracle as O

KEYS = ('embed', 'attrs', 'masks', 'sep_pwr')
BAR = 1e-4
BAR_LOGITS = 1e-5


def _err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d = a - b
    return dict(max_rel=float(np.abs(d).max() / (np.abs(b).max() + 1e-300)),
                mse=float((d * d).mean()),
                rms_rel=float(np.sqrt((d * d).mean()) / (np.sqrt((b * b).mean()) + 1e-300)))


def oracle_pair(src, params, cfg):
    '''(float64 oracle outputs, float32 oracle outputs) of model_forward at `params`'''
    r64 = O.model_forward(np.asarray(src).astype(np.complex128), params, cfg, dtype=np.float64)
    r32 = O.model_forward(np.asarray(src).astype(np.complex64), params, cfg, dtype=np.float32)
    return r64, r32


def torch_f32(src, params, cfg, keys):
    '''the second restatement (oracle/torch_ref.py: hoisted input product, torch-CPU
    kernels, i.e. another summation order) evaluated in float32'''
    import torch
    from . import torch_ref as R
    tp = {k: torch.tensor(np.asarray(v), dtype=torch.float32) for k, v in params.items()}
    with torch.no_grad():
        r = R.model_forward(torch.tensor(np.asarray(src).astype(np.complex64)), tp, cfg)
    return {k: r[k].numpy() for k in keys if k in r}


def _logits(r):
    '''embed_flat . attr^T (app/modules.py:587-589) in float64 from a path's own tensors'''
    e = np.asarray(r['embed'], dtype=np.float64)
    a = np.asarray(r['attrs'], dtype=np.float64)
    B, C, E = a.shape
    return np.einsum('bne,bce->bnc', e.reshape(B, -1, E), a)


def _logits_entry(got, r64, r32, t32):
    ref = _logits(r64)
    e_hip = _err(_logits(got), ref)
    e_f32 = _err(_logits(r32), ref)
    if 'embed' in t32 and 'attrs' in t32:
        e_t32 = _err(_logits(t32), ref)
        if e_t32['max_rel'] > e_f32['max_rel']:
            e_f32 = e_t32
    bound = max(BAR_LOGITS, 2.0 * e_f32['max_rel'])
    return dict(hip_vs_f64=e_hip, f32_vs_f64=e_f32, bound=bound, ok=bool(e_hip['max_rel'] <= bound),
                max_abs_logit=float(np.abs(ref).max()))


def parity_report(got, src, params, cfg, keys=KEYS):
    '''got: {key: ndarray} product outputs for the mixtures in `src` (masks as
    [B,T,F,C]).  Returns {key: {hip_vs_f64: {max_rel, rms_rel}, f32_vs_f64: {...},
    bound, ok}, ..., ok: bool, subset_choice_equal: bool}.'''
    r64, r32 = oracle_pair(src, params, cfg)
    t32 = torch_f32(src, params, cfg, keys)
    rep = {}
    ok = True
    for k in keys:
        if k not in got:
            continue
        e_hip = _err(got[k], r64[k])
        # noise floor of float32 arithmetic: the larger of two float32 evaluations of the
        # reference's arithmetic with different summation orders (numpy per-timestep
        # [x,h]W restatement; torch restatement with the hoisted input product)
        e_f32 = _err(r32[k], r64[k])
        e_t32 = _err(t32[k], r64[k]) if k in t32 else e_f32
        if e_t32['max_rel'] > e_f32['max_rel']:
            e_f32 = dict(e_t32, which='torch_ref float32')
        else:
            e_f32 = dict(e_f32, which='danet_oracle float32', other=e_t32['max_rel'])
        bound = max(BAR, 2.0 * e_f32['max_rel'])
        rep[k] = dict(hip_vs_f64=e_hip, f32_vs_f64=e_f32, bound=bound,
                      ok=bool(e_hip['max_rel'] <= bound))
        ok = ok and rep[k]['ok']
    if 'embed' in got and 'attrs' in got:
        rep['logits'] = _logits_entry(got, r64, r32, t32)
        ok = ok and rep['logits']['ok']
    rep['perm_idx_equal_f32_f64'] = bool(np.array_equal(r32['perm_idx'], r64['perm_idx']))
    if 'perm_idx' in got:
        rep['perm_idx_equal'] = bool(np.array_equal(np.asarray(got['perm_idx']), r64['perm_idx']))
        ok = ok and rep['perm_idx_equal']
    rep['ok'] = bool(ok)
    return rep
