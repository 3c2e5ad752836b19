#!/usr/bin/env python
'''The four streaming kernels of the train step's heads (anchor / truth estimator forward, fused
separator + PIT forward, its backward (dattr only), estimator backward with the separator term
recomputed), each timed in a loop through the C ABI.  python tools/bench_heads_fused.py [--cfg4]'''
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.load_package()
from danet_amd import _lib  # noqa: E402


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    dev = torch.device('cuda')
    cfg4 = '--cfg4' in sys.argv
    B, T, F = 32, 128, 129
    E, C, A = (40, 3, 6) if cfg4 else (20, 2, 6)
    N = T * F
    torch.manual_seed(0)
    L, p, st = _lib.load(), _lib.ptr, _lib.stream
    embed = torch.randn(B, N, E, device=dev) * 0.5
    anchors = torch.randn(A, E, device=dev) * 0.5
    mix = torch.rand(B, N, device=dev) * 100
    src = torch.randn(B, C, N, 2, device=dev) * 50
    phasor = torch.randn(B, N, 2, device=dev)
    phasor = phasor / phasor.norm(dim=-1, keepdim=True)
    P = 1
    for i in range(C):
        P = P * (A - i) // (i + 1)
    attr = torch.empty(B, C, E, device=dev)
    asets = torch.empty(B, P, C, E, device=dev)
    asum = torch.empty(B, P, C, device=dev)
    choice = torch.empty(B, dtype=torch.int32, device=dev)
    wn = _lib.ws_bytes(_lib.WS_ATTRACTOR_ANCHOR, B, C, N, E, A)
    ws = torch.empty(wn, dtype=torch.uint8, device=dev)
    rec = torch.empty(_lib.ws_bytes(_lib.WS_SEPARATE_PIT_RECORDS, B, N), dtype=torch.uint8, device=dev)
    wn2 = _lib.ws_bytes(_lib.WS_SEPARATE_PIT, B, C, N, E)
    ws2 = torch.empty(wn2, dtype=torch.uint8, device=dev)
    dattr = torch.empty(B, C, E, device=dev)
    dembed = torch.empty(B, N, E, device=dev)
    mb = embed.numel() * 4 / 1e6
    small = (mix.numel() + src.numel() + phasor.numel()) * 4 / 1e6

    def chk(rc):
        assert rc == 0, L.danet_last_error()

    def a_fwd():
        chk(L.danet_attractor_anchor_fwd(st(), B, C, N, E, A, p(embed), p(anchors), p(attr), p(asets),
                                         p(asum), p(choice), p(ws), wn))

    def s_fwd():
        chk(L.danet_separate_pit_fwd_records(st(), 0, 0, B, C, N, E, p(mix), p(attr), p(embed), p(src),
                                             p(phasor), None, p(rec), None))

    # round 6: the forward also leaves the attractor-gradient partials of every permutation (C <= 2)
    gn = _lib.ws_bytes(_lib.WS_SEPARATE_PIT_GRAD, B, C, N, E)
    gpart = torch.empty(max(gn, 4), dtype=torch.uint8, device=dev)

    def s_fwd_g():
        chk(L.danet_separate_pit_fwd_records(st(), 0, 0, B, C, N, E, p(mix), p(attr), p(embed), p(src),
                                             p(phasor), None, p(rec), p(gpart)))

    def a_bwd_g():
        chk(L.danet_attractor_anchor_bwd_embed_sep(st(), B, C, N, E, A, None, p(embed), p(anchors),
                                                   p(attr), p(asum), p(choice), 0, 0, p(mix), p(src),
                                                   p(phasor), None, p(rec), 1.0, None, p(dembed), p(ws), wn,
                                                   p(gpart)))

    def s_bwd():
        chk(L.danet_separate_pit_bwd(st(), 0, 0, B, C, N, E, p(mix), p(attr), p(embed), p(src), p(phasor),
                                     None, p(rec), 1.0, None, None, p(dattr), p(ws2), wn2))

    def a_bwd():
        chk(L.danet_attractor_anchor_bwd_embed_sep(st(), B, C, N, E, A, p(dattr), p(embed), p(anchors),
                                                   p(attr), p(asum), p(choice), 0, 0, p(mix), p(src),
                                                   p(phasor), None, p(rec), 1.0, None, p(dembed), p(ws), wn, None))

    for _ in range(30):
        a_fwd()
    rows = [('anchor_fwd (+final)', a_fwd, mb), ('sep_pit_fwd', s_fwd, mb + small),
            ('sep_pit_bwd (dattr only, + chunk sum)', s_bwd, mb + small),
            ('anchor_bwd_embed_sep', a_bwd, 2 * mb + small)]
    tot = 0.0
    for name, fn, mbs in rows:
        us = timeit(fn)
        tot += us
        print('%-40s %7.1f us   %.2f TB/s on %.0f MB' % (name, us, mbs / us, mbs))

    def chain():
        a_fwd(); s_fwd(); s_bwd(); a_bwd()
    print('%-40s %7.1f us   (sum of the four: %.1f)' % ('chain', timeit(chain), tot))
    if gn:
        tot_g = 0.0
        for name, fn, mbs in [('sep_pit_fwd + gradient partials', s_fwd_g, mb + small),
                              ('anchor_bwd_embed_sep from the partials', a_bwd_g, 2 * mb + small)]:
            us = timeit(fn)
            tot_g += us
            print('%-40s %7.1f us   %.2f TB/s on %.0f MB' % (name, us, mbs / us, mbs))

        def chain_g():
            a_fwd(); s_fwd_g(); a_bwd_g()
        print('%-40s %7.1f us   (round 6: three launches)' % ('chain, gradient partials in the forward', timeit(chain_g)))
        s_fwd(); s_bwd(); a_bwd()
        ref = dembed.clone()
        s_fwd_g(); a_bwd_g()
        torch.cuda.synchronize()
        print('dembed vs the four-launch chain: max |diff| / max |ref| = %.2e' % float((ref - dembed).abs().max() / ref.abs().max()))


if __name__ == '__main__':
    main()
