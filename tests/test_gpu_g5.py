'''
HIP path vs the COMMITTED G5 fixtures (run with -m gpu).

tests/golden/oracle_g5_*.npz are oracle-generated (NOT reference-generated -- the
reference's model path needs TF1) and frozen: tests/test_oracle_cpu.py checks
the live oracle against them, this file checks the product against the same committed
arrays, so the product is never compared only with an oracle edited in the same commit.
Keys = the reference's debug_fetches (main.py:389-397, app/modules.py:540-543,571,600)
+ loss / SNR / perm idx + every parameter gradient.
'''
import os

import numpy as np
import pytest
import torch

from oracle import g5

pytestmark = pytest.mark.gpu
TOL = 1e-4
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.mark.parametrize('name', sorted(g5.CASES))
def test_hip_path_matches_committed_g5(hp, name):
    from danet_amd.model import Model
    from danet_amd import ops
    src, p, cfg = g5.case_inputs(name)
    B, C = src.shape[:2]
    hp.load(dict(BATCH_SIZE=B, MAX_N_SIGNAL=C, FFT_SIZE=cfg['FFT'], FFT_STRIDE=max(cfg['FFT'] // 4, 1),
                 EMBED_SIZE=cfg['E'], NUM_LSTM_LAYERS=cfg['L'], LSTM_HDIM=cfg['H'],
                 NUM_ANCHOR=cfg['A'], ENCODER_TYPE='bilstm-orig',
                 TRAIN_ESTIMATOR_METHOD=cfg['train_est'], INFER_ESTIMATOR_METHOD=cfg['infer_est'],
                 SEPARATOR_TYPE=cfg['separator'], DEBUG=True))
    hp.digest()
    model = Model('g5', device='cuda').build()
    assert set(model.vars) == set(p), (sorted(model.vars), sorted(p))
    model.load_param_dict(p)
    s = torch.as_tensor(src).cuda()
    model._flat_grad.zero_()
    out = model.forward(s, with_valid=True)
    out['loss'].backward()
    dbg = {}
    for mod in (model.estimator, model.valid_estimator, model.separator):
        dbg.update(getattr(mod, 'debug_fetches', {}) or {})
    got = dict(
        embed=out['embed'], attrs=out['attrs'], sep_pwr=out['sep_pwr'], loss=out['loss'],
        SNR=out['SNR'], perm_idx=out['perm_idx'], valid_attrs=out['valid_attrs'],
        sep_pwr_valid=out['sep_pwr_valid'], valid_loss=out['valid_loss'],
        valid_SNR=out['valid_SNR'], valid_perm_idx=out['valid_perm_idx'],
        output=ops.reattach_phase(out['sep_pwr'].detach(), out['phasor'], out['perm_idx']))
    if 'asets' in dbg:                       # the anchor estimator's fetches (last one called)
        got['asets'], got['subset_choice'] = dbg['asets'], dbg['subset_choice']
    got = {k: v.detach().cpu().numpy() for k, v in got.items()}
    # masks: separator debug fetch holds the LAST call (valid branch); recompute the train one
    _, masks = ops.SeparateFn.apply(out['mix_pwr'], out['attrs'].detach(),
                                    out['embed'].detach().reshape(B, -1, cfg['E']),
                                    0 if cfg['separator'] == 'dot-softmax-orig' else 1, True)
    got['masks'] = masks.cpu().numpy()
    for k, g in model.grad_dict().items():
        got['grad:' + k] = g
    torch.cuda.synchronize()
    assert ops.lstm_status_ok()
    fix = dict(np.load(os.path.join(GOLD, 'oracle_g5_%s.npz' % name)))
    fix.pop('_label')
    errs = g5.compare(fix, got, TOL)
    need = {'embed', 'attrs', 'masks', 'asets', 'subset_choice', 'output', 'loss', 'SNR', 'perm_idx'}
    assert need <= {k.split('__')[0] for k in errs}, need - {k.split('__')[0] for k in errs}
    bad = {k: e for k, e in errs.items()
           if not e <= (2 * TOL if k.startswith('grad:') else TOL)}
    assert not bad, bad
