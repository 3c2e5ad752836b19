'''
CPU ORACLE (second restatement) -- TEST INFRASTRUCTURE ONLY.

Independent torch-CPU autograd restatement of the reference's model path, used
(a) to cross-check oracle/danet_oracle.py in float64 (two restatements must
agree, SURVEY 7 "No TF oracle"), (b) as the gradient authority for every HIP
`*_bwd` kernel (itself checked by float64 finite differences, K11), and
(c) as bench.py's `cpu_baseline` ("port": TF1 cannot run anywhere here).

PARITY UNPINNED by the reference for the model path (it ships no tests; TF1 is
not installable) -- see oracle/danet_oracle.py header.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  Reference citations are file:line under /root/reference.
'''
import itertools
import math

import torch


def frontend(src):
    '''main.py:233-240.  src complex[B,C,T,F]'''
    mix = src.sum(dim=1)
    src_pwr = src.abs()
    phase = torch.atan2(mix.imag, mix.real)
    mix_pwr = mix.abs()
    mix_log = torch.log1p(mix_pwr)
    return dict(mix=mix, src_pwr=src_pwr, phase=phase, mix_pwr=mix_pwr,
                mix_log=mix_log)


def lstm_scan(x, W, b, H, reverse=False):
    '''main.py:76-132 + app/ops.py:110-148 (g linear; order g,i,f,o).
    x[B,T,D] -> [B,T,H].  `reverse` = run on x[:, ::-1] and flip back
    (app/modules.py:132-136).'''
    B, T, D = x.shape
    c = x.new_zeros(B, H)
    h = x.new_zeros(B, H)
    Wx, Wh = W[:D], W[D:]
    gx = x @ Wx + b                      # hoisted input half; same sum order
    outs = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        a = gx[:, t] + h @ Wh
        g = a[:, :H]
        i = torch.sigmoid(a[:, H:2 * H])
        f = torch.sigmoid(a[:, 2 * H:3 * H])
        o = torch.sigmoid(a[:, 3 * H:])
        c = i * g + f * c
        h = o * torch.tanh(c)
        outs[t] = h
    return torch.stack(outs, dim=1)


def bilstm_encoder(x, params, H, L, E):
    '''app/modules.py:207-260'''
    B, T, F = x.shape
    x = x - x.mean(dim=(1, 2), keepdim=True)
    for l in range(L):
        fwd = lstm_scan(x, params['global/encoder/lstm%d_fwd/LSTM/linear/W' % l],
                        params['global/encoder/lstm%d_fwd/LSTM/linear/B' % l], H)
        bwd = lstm_scan(x, params['global/encoder/lstm%d_bwd/LSTM/linear/W' % l],
                        params['global/encoder/lstm%d_bwd/LSTM/linear/B' % l], H,
                        reverse=True)
        x = torch.cat([fwd, bwd], dim=-1)
    y = x - x.mean(dim=(1, 2), keepdim=True)
    out = y @ params['global/encoder/output/W']
    return out.reshape(B, T, F, E)


def lstm_encoder(x, params, H, L, E):
    '''app/modules.py:148-196'''
    B, T, F = x.shape
    x = x - x.mean(dim=(1, 2), keepdim=True)
    for l in range(L):
        x = lstm_scan(x, params['global/encoder/lstm%d/LSTM/linear/W' % l],
                      params['global/encoder/lstm%d/LSTM/linear/B' % l], H)
    y = x - x.mean(dim=(1, 2), keepdim=True)
    out = y @ params['global/encoder/output/W']
    return out.reshape(B, T, F, E)


def toy_encoder(x, params, E, relu_leak=0.):
    '''ToyEncoder (app/modules.py:96-116): linear -> (leaky) relu (app/ops.py:93-107) ->
    linear, both with bias (ops.lyr_linear, app/ops.py:72-89)'''
    B, T, F = x.shape
    m = x @ params['global/encoder/linear0/W'] + params['global/encoder/linear0/B']
    m = torch.relu(m) if relu_leak == 0. else torch.maximum(m * relu_leak, m)
    o = m @ params['global/encoder/linear1/W'] + params['global/encoder/linear1/B']
    return o.reshape(B, T, F, E)


def _truth_family(embed, src_pwr, wgt, denom_add):
    '''app/modules.py:390-487: segment sums by argmax_c |src|'''
    B, T, F, E = embed.shape
    C = src_pwr.shape[1]
    ef = embed.reshape(B, -1, E)
    idx = src_pwr.argmax(dim=1).reshape(B, -1)            # ties -> lowest index
    onehot = torch.nn.functional.one_hot(idx, C).to(embed.dtype)   # [B,N,C]
    w = wgt.reshape(B, -1, 1)
    num = torch.einsum('bnc,bne->bce', onehot * w, ef)
    den = (onehot * w).sum(dim=1)                          # [B,C]
    return num / (den[..., None] + denom_add)


def est_truth(embed, src_pwr, mix_pwr=None, eps=None):
    return _truth_family(embed, src_pwr,
                         torch.ones_like(embed[..., 0]), 1.0)


def est_truth_threshold(embed, src_pwr, mix_pwr, eps=1e-7):
    return _truth_family(embed, src_pwr, (mix_pwr > 5.0).to(embed.dtype), eps)


def est_truth_weighted(embed, src_pwr, mix_pwr, eps=1e-7):
    return _truth_family(embed, src_pwr, mix_pwr, eps)


def est_anchor(embed, anchors, C, return_all=False):
    '''app/modules.py:501-545'''
    B = embed.shape[0]
    A = anchors.shape[0]
    combs = torch.tensor(list(itertools.combinations(range(A), C)))
    sets = anchors[combs]
    logit = torch.einsum('btfe,pce->bptfc', embed, sets)
    assign = torch.softmax(logit, dim=-1)
    asets = torch.einsum('bptfc,btfe->bpce', assign, embed)
    asets = asets / assign.sum(dim=(2, 3))[..., None]
    gram = asets @ asets.transpose(-1, -2)
    sim = gram.amax(dim=(-1, -2))
    choice = sim.argmin(dim=1)
    attr = asets[torch.arange(B), choice]
    if return_all:
        return attr, dict(asets=asets, subset_choice=choice, sim=sim)
    return attr


def est_kmeans(embed, anchors, C, mix_pwr=None, iters=10, eps=1e-7):
    '''k-means attractor estimation (BASELINE cfg 5).  NOT in the reference (README.md:216) --
    this restates what the product's extension documents, from the reference's own pieces:
    start at the anchor estimator's attractors (app/modules.py:501-545), then `iters` Lloyd
    iterations: assign every bin to the attractor with the largest dot product (the
    separator's similarity, app/modules.py:587-589; ties -> lowest index like tf.argmax) and
    recompute each attractor as the |mix|-weighted mean of its bins (the truth-weighted
    formula, app/modules.py:476-482, with estimated assignments).'''
    B, T, F, E = embed.shape
    attr = est_anchor(embed, anchors, C)
    w = torch.ones_like(embed[..., 0]) if mix_pwr is None else mix_pwr
    ef = embed.reshape(B, -1, E)
    for _ in range(int(iters)):
        idx = (ef @ attr.transpose(1, 2)).argmax(dim=-1)            # [B, N]
        onehot = torch.nn.functional.one_hot(idx, C).to(embed.dtype) * w.reshape(B, -1, 1)
        attr = torch.einsum('bnc,bne->bce', onehot, ef) / (onehot.sum(dim=1)[..., None] + eps)
    return attr


def infer_forward(mix, params, cfg):
    '''the inference fetches (main.py:384-385, :333-335, :685-690): complex mixture [B,T,F] ->
    dict(embed, attrs, masks, sep_pwr, sep) with the inference estimator and the mixture
    phase.  cfg['infer_est'] in ('anchor', 'kmeans' (extension)).'''
    fe = frontend(mix[:, None])
    H, L, E, C = cfg['H'], cfg['L'], cfg['E'], cfg['C']
    embed = bilstm_encoder(fe['mix_log'], params, H, L, E)
    B = embed.shape[0]
    anchors = params['global/infer_estimator/anchors']
    if cfg['infer_est'] == 'kmeans':
        attrs = est_kmeans(embed, anchors, C, fe['mix_pwr'], cfg.get('kmeans_iters', 10),
                           cfg.get('eps', 1e-7))
    else:
        attrs = est_anchor(embed, anchors, C)
    act = {'dot-softmax-orig': 'softmax', 'dot-sigmoid-orig': 'sigmoid'}[cfg['separator']]
    sep_pwr, masks = sep_dot(fe['mix_pwr'], attrs, embed.reshape(B, -1, E), act)
    ph = fe['phase'][:, None]
    sep = torch.complex(torch.cos(ph) * sep_pwr, torch.sin(ph) * sep_pwr)
    return dict(embed=embed, attrs=attrs, masks=masks, sep_pwr=sep_pwr, sep=sep)


def sep_dot(mix_pwr, attr, embed_flat, act):
    '''app/modules.py:556-603'''
    B, T, F = mix_pwr.shape
    C = attr.shape[1]
    logits = (embed_flat @ attr.transpose(1, 2)).reshape(B, T, F, C)
    masks = torch.softmax(logits, dim=-1) if act == 'softmax' \
        else torch.sigmoid(logits)
    return (mix_pwr[..., None] * masks).permute(0, 3, 1, 2), masks


def pit_mse_loss(x, y):
    '''app/ops.py:374-431'''
    B, C = x.shape[:2]
    perms = torch.tensor(list(itertools.permutations(range(C))))
    d = x[:, :, None] - y[:, None, :]
    if x.is_complex() and y.is_complex():
        sq = d.real ** 2 + d.imag ** 2
    else:
        sq = d ** 2
    cross = sq.mean(dim=(3, 4))
    loss_sets = torch.stack(
        [cross[:, torch.arange(C), perm].sum(dim=1) for perm in perms], dim=1)
    idx = loss_sets.argmin(dim=1)
    loss = loss_sets[torch.arange(B), idx].mean()
    return loss, perms, idx


def batch_snr(clear, noisy, eps=1e-7):
    '''app/ops.py:191-222'''
    noise = clear - noisy
    if clear.is_complex():
        clear, noise = clear.abs(), noise.abs()
    axes = tuple(range(1, clear.dim()))
    sp = (clear ** 2).mean(dim=axes)
    npw = (noise ** 2).mean(dim=axes)
    return 4.342944819 * (torch.log(sp + eps) - torch.log(npw + eps))


def model_forward(src, params, cfg):
    '''main.py:208-337 (train branch + valid branch).  params: dict of tensors
    (requires_grad as the caller wishes).'''
    fe = frontend(src)
    H, L, E, C = cfg['H'], cfg['L'], cfg['E'], cfg['C']
    eps = cfg.get('eps', 1e-7)
    if cfg.get('encoder', 'bilstm-orig') == 'bilstm-orig':
        embed = bilstm_encoder(fe['mix_log'], params, H, L, E)
    elif cfg['encoder'] == 'toy':
        embed = toy_encoder(fe['mix_log'], params, E, cfg.get('relu_leak', 0.))
    else:
        embed = lstm_encoder(fe['mix_log'], params, H, L, E)
    B, T, F, _ = embed.shape
    ef = embed.reshape(B, -1, E)

    def run_est(name, scope):
        if name == 'anchor':
            return est_anchor(embed, params['global/%s/anchors' % scope], C)
        fn = {'truth': est_truth, 'truth-threshold': est_truth_threshold,
              'truth-weighted': est_truth_weighted}[name]
        return fn(embed, fe['src_pwr'], fe['mix_pwr'], eps)

    attrs = run_est(cfg['train_est'], 'train_estimator')
    act = {'dot-softmax-orig': 'softmax', 'dot-sigmoid-orig': 'sigmoid'}[
        cfg['separator']]
    sep_pwr, masks = sep_dot(fe['mix_pwr'], attrs, ef, act)
    ph = fe['phase'][:, None]
    sep = torch.complex(torch.cos(ph) * sep_pwr, torch.sin(ph) * sep_pwr)
    loss, perms, idx = pit_mse_loss(src, sep)
    sep_perm = sep[torch.arange(B)[:, None], perms[idx]]
    snr = batch_snr(src, sep_perm, eps).mean()
    out = dict(fe, embed=embed, attrs=attrs, masks=masks, sep_pwr=sep_pwr,
               sep=sep, loss=loss, perm_idx=idx, SNR=snr)
    if cfg.get('with_valid', False):
        if cfg['infer_est'] == cfg['train_est']:
            vattrs = attrs
        else:
            vattrs = run_est(cfg['infer_est'], 'infer_estimator')
        sep_pwr_v, _ = sep_dot(fe['mix_pwr'], vattrs, ef, act)
        vloss, _, vidx = pit_mse_loss(fe['src_pwr'], sep_pwr_v)
        out.update(valid_attrs=vattrs, sep_pwr_valid=sep_pwr_v,
                   valid_loss=vloss, valid_perm_idx=vidx)
    return out


def tf_adam_step_(params, grads, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-8,
                  clip=100.0):
    '''in-place TF1 Adam with value clip (main.py:359-363, app/ozers.py:15-18)'''
    lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    with torch.no_grad():
        for k in params:
            g = grads[k]
            if g is None:
                continue
            if clip is not None:
                g = g.clamp(-clip, clip)
            m[k].mul_(b1).add_(g, alpha=1 - b1)
            v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
            params[k].sub_(lr_t * m[k] / (v[k].sqrt() + eps))
