#!/usr/bin/env python
'''Per-step phase timeline of the persistent LSTM kernels (diagnostic).

Build the trace variant first:   python danet-tensorflow_amd/_build.py --trace
Run on a GPU box:                python tools/trace_lstm.py [B T H]

Workgroup 0 / wave 0 stamps each step with the 100 MHz wall clock:
  A top of step | B exchange valid (retries in G) | C MFMAs done | D barrier 1 |
  E h/da published (stores issued) | F barrier 2
'''
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

g.load_package()
from danet_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, 'danet-tensorflow_amd', 'csrc', 'libdanet_hip_trace.so')
L = _lib.load()
ptr = _lib.ptr


def report(name, ws, T):
    tr = ws[64:64 + T * 64].view(torch.int64).view(T, 8).cpu().numpy().astype(np.float64)
    tr = tr[2:]                                    # skip the cold first steps
    us = lambda a: a / 100.0                       # 100 MHz ticks -> us
    step = us(np.diff(tr[:, 0]))
    ph = dict(A_to_B_exchange=us(tr[:, 1] - tr[:, 0]), B_to_C_mfma=us(tr[:, 2] - tr[:, 1]),
              C_to_D_reduce_barrier=us(tr[:, 3] - tr[:, 2]), D_to_E_gates_publish=us(tr[:, 4] - tr[:, 3]),
              E_to_F_barrier=us(tr[:, 5] - tr[:, 4]))
    print('%s: step %.2f us (median %.2f)  retries/step %.2f' % (
        name, step.mean(), np.median(step), tr[:, 6].mean()))
    for k, v in ph.items():
        print('   %-24s mean %.2f  median %.2f  p90 %.2f' % (k, v.mean(), np.median(v), np.percentile(v, 90)))


def report_fused(name, ws, T):
    '''lstm_fwd_fx_kernel: 0 top | 1 first quarter of the input half + loads issued | 2 rest
    of the input half issued | 3 exchange valid (6 retries) | 4 recurrent MFMAs + reduce barrier
    | 5 gate math, publish | 7 barrier'''
    tr = ws[64:64 + T * 64].view(torch.int64).view(T, 8).cpu().numpy().astype(np.float64)
    tr = tr[2:]
    us = lambda a: a / 100.0
    step = us(np.diff(tr[:, 0]))
    ph = dict(top_to_loads=us(tr[:, 1] - tr[:, 0]), input_half_rest=us(tr[:, 2] - tr[:, 1]),
              rest_to_valid=us(tr[:, 3] - tr[:, 2]), recurrent_reduce=us(tr[:, 4] - tr[:, 3]),
              gate_publish=us(tr[:, 5] - tr[:, 4]), barrier=us(tr[:, 7] - tr[:, 5]),
              to_next_top=us(tr[1:, 0] - tr[:-1, 7]))
    print('%s: step %.2f us (median %.2f)  retries/step %.2f' % (
        name, step.mean(), np.median(step), tr[:, 6].mean()))
    for k, v in ph.items():
        print('   %-24s mean %.2f  median %.2f  p90 %.2f' % (k, v.mean(), np.median(v), np.percentile(v, 90)))


def fused(B, T, H, D):
    dev = torch.device('cuda')
    st = torch.cuda.current_stream().cuda_stream
    ndir = 2
    if L.danet_lstm_fwd_fused_supported(T, B, H, ndir, D) != 1:
        print('fused forward: shape outside the envelope')
        return
    x = torch.randn(T, B, D, device=dev) * 0.5
    W = [torch.randn(D + H, 4 * H, device=dev) * (0.75 / H ** 0.5) for _ in range(2)]
    b = [torch.zeros(4 * H, device=dev) for _ in range(2)]
    ypad = torch.empty(T + 2, B, 2 * H, device=dev)
    gates = [torch.empty(T * B, 4 * H, device=dev) for _ in range(2)]
    cells = [torch.empty(T * B, H, device=dev) for _ in range(2)]
    n = _lib.ws_bytes(_lib.WS_LSTM, T, B, H, ndir)
    for it in range(3):
        ws = torch.zeros(n, dtype=torch.uint8, device=dev)
        _lib.check(L.danet_lstm_fwd_fused(st, T, B, H, ndir, ptr(x), D, D, ptr(W[0]), ptr(W[1]), 4 * H,
                                          ptr(b[0]), ptr(b[1]), ptr(ypad), 2 * H, ptr(gates[0]),
                                          ptr(gates[1]), ptr(cells[0]), ptr(cells[1]), ptr(ws), n, None, 0))
        torch.cuda.synchronize()
    assert int(ws[:4].view(torch.int32)[0]) == 0
    report_fused('lstm_fwd_fused D=%d' % D, ws, T)


def main():
    B, T, H = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (32, 128, 300)
    if _lib.get_option('lstm_fwd_fused') == 1:
        fused(B, T, H, 2 * H)
        fused(B, T, H, 132)
    dev = torch.device('cuda')
    st = torch.cuda.current_stream().cuda_stream
    ndir = 2
    gx = [torch.randn(T * B, 4 * H, device=dev) * 0.5 for _ in range(2)]
    Wh = [torch.randn(H, 4 * H, device=dev) * (0.75 / H ** 0.5) for _ in range(2)]
    ypad = torch.empty(T + 2, B, 2 * H, device=dev)
    cells = [torch.empty(T * B, H, device=dev) for _ in range(2)]
    n = _lib.ws_bytes(_lib.WS_LSTM, T, B, H, ndir)
    for it in range(3):
        ws = torch.zeros(n, dtype=torch.uint8, device=dev)
        gates = [x.clone() for x in gx]
        _lib.check(L.danet_lstm_fwd(st, T, B, H, ndir, ptr(gates[0]), ptr(gates[1]), ptr(Wh[0]),
                                    ptr(Wh[1]), 4 * H, ptr(ypad), 2 * H, ptr(gates[0]),
                                    ptr(gates[1]), ptr(cells[0]), ptr(cells[1]), ptr(ws), n, None, 0))
        torch.cuda.synchronize()
    assert int(ws[:4].view(torch.int32)[0]) == 0
    report('lstm_fwd', ws, T)
    dy = torch.randn(T, B, 2 * H, device=dev)
    das = [torch.empty(T * B, 4 * H, device=dev) for _ in range(2)]
    for it in range(3):
        ws = torch.zeros(n, dtype=torch.uint8, device=dev)
        _lib.check(L.danet_lstm_bwd(st, T, B, H, ndir, ptr(dy), 2 * H, ptr(Wh[0]), ptr(Wh[1]), 4 * H,
                                    ptr(gates[0]), ptr(gates[1]), ptr(cells[0]), ptr(cells[1]),
                                    ptr(das[0]), ptr(das[1]), None, None, 0.0, ptr(ws), n, None, 0))
        torch.cuda.synchronize()
    assert int(ws[:4].view(torch.int32)[0]) == 0
    report('lstm_bwd', ws, T)
    # the same BPTT launch with a layer's weight-gradient group (4 products, K = T*B, capped to one
    # workgroup per CU, as in the train step) running beside it on another stream: which phase
    # pays for the company?
    from danet_amd import ops
    D = 2 * H
    x = torch.randn(T * B, D, device=dev)
    hp = torch.randn(T * B + 2 * B, 2 * H, device=dev)
    dWs = [torch.empty(D + H, 4 * H, device=dev) for _ in range(2)]
    side = torch.cuda.Stream()

    def group():
        probs = []
        for d in range(2):
            probs.append((x, D, das[d], 4 * H, dWs[d], 4 * H, D, 4 * H, 0.0))
            probs.append((hp, 2 * H, das[d], 4 * H, dWs[d][D:], 4 * H, H, 4 * H, 0.0))
        ops.gemm_group(probs, T * B, transA=True, max_workgroups=256)

    for variant in ('beside the capped group',):
        for it in range(3):
            ws = torch.zeros(n, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            with torch.cuda.stream(side):
                group()
            _lib.check(L.danet_lstm_bwd(st, T, B, H, ndir, ptr(dy), 2 * H, ptr(Wh[0]), ptr(Wh[1]), 4 * H,
                                        ptr(gates[0]), ptr(gates[1]), ptr(cells[0]), ptr(cells[1]),
                                        ptr(das[0]), ptr(das[1]), None, None, 0.0, ptr(ws), n, None, 0))
            torch.cuda.synchronize()
        assert int(ws[:4].view(torch.int32)[0]) == 0
        report('lstm_bwd ' + variant, ws, T)


if __name__ == '__main__':
    main()
