#!/usr/bin/env python
'''A/B the persistent LSTM kernels' placement / publish modes (GPU box).

    python tools/lstm_modes.py [B T H]

For each BPTT geometry (library options lstm_bwd_u / lstm_bwd_s / lstm_xmap) it times danet_lstm_fwd / _bwd
standalone and checks the outputs bit-exactly against the default mode.
'''
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

g.load_package()
from danet_amd import _lib  # noqa: E402

L = _lib.load()
ptr = _lib.ptr


def main():
    B, T, H = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (32, 128, 300)
    dev = torch.device('cuda')
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(0)
    gx = [torch.randn(T * B, 4 * H, device=dev) * 0.5 for _ in range(2)]
    Wh = [torch.randn(H, 4 * H, device=dev) * (0.75 / H ** 0.5) for _ in range(2)]
    dy = torch.randn(T, B, 2 * H, device=dev)
    n = _lib.ws_bytes(_lib.WS_LSTM, T, B, H, 2)
    ws = torch.zeros(n, dtype=torch.uint8, device=dev)

    def fwd():
        ypad = torch.empty(T + 2, B, 2 * H, device=dev)
        gates = [x.clone() for x in gx]
        cells = [torch.empty(T * B, H, device=dev) for _ in range(2)]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(L.danet_lstm_fwd(st, T, B, H, 2, ptr(gates[0]), ptr(gates[1]), ptr(Wh[0]), ptr(Wh[1]),
                                    4 * H, ptr(ypad), 2 * H, ptr(gates[0]), ptr(gates[1]),
                                    ptr(cells[0]), ptr(cells[1]), ptr(ws), n, None, 0))
        e1.record()
        torch.cuda.synchronize()
        assert int(ws[:4].view(torch.int32)[0]) == 0, 'status'
        return e0.elapsed_time(e1) * 1e3, ypad, gates, cells

    def bwd(gates, cells):
        das = [torch.empty(T * B, 4 * H, device=dev) for _ in range(2)]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(L.danet_lstm_bwd(st, T, B, H, 2, ptr(dy), 2 * H, ptr(Wh[0]), ptr(Wh[1]), 4 * H,
                                    ptr(gates[0]), ptr(gates[1]), ptr(cells[0]), ptr(cells[1]),
                                    ptr(das[0]), ptr(das[1]), None, None, 0.0, ptr(ws), n, None, 0))
        e1.record()
        torch.cuda.synchronize()
        assert int(ws[:4].view(torch.int32)[0]) == 0, 'status'
        return e0.elapsed_time(e1) * 1e3, das

    ref = None
    modes = [dict()]
    for u, ss in (('32', ('4', '5', '6')), ('16', ('2', '3')), ('8', ('1',))):
        for sv in ss:
            modes.append(dict(DANET_LSTM_BWD_U=u, DANET_LSTM_BWD_S=sv))
    modes.append(dict(DANET_LSTM_BWD_U='32', DANET_LSTM_BWD_S='5', DANET_LSTM_XMAP='0'))
    for m in modes:
        _lib.apply_env_options()          # defaults (+ process environment), then this mode
        for k, v in m.items():
            _lib.set_option(k[len('DANET_'):].lower(), int(v))
        tf, tb = [], []
        for it in range(12):
            a, ypad, gates, cells = fwd()
            b, das = bwd(gates, cells)
            tf.append(a); tb.append(b)
        tf, tb = sorted(tf[2:]), sorted(tb[2:])
        out = (ypad[1:-1].clone(), das[0].clone(), das[1].clone())
        err = 0.0
        if ref is None:
            ref = out
        else:
            err = max(float((x - y).abs().max() / y.abs().max()) for x, y in zip(out, ref))
        print('%-62s fwd %.1f us  bwd %.1f us (min %.1f)  per-step %.2f / %.2f us  max rel diff vs first %.2e'
              % (m, tf[len(tf) // 2], tb[len(tb) // 2], tb[0],
                 tf[len(tf) // 2] / T, tb[len(tb) // 2] / T, err), flush=True)


if __name__ == '__main__':
    main()
