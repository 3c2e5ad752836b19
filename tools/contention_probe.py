#!/usr/bin/env python
'''What slows the persistent BPTT kernel when another kernel shares the GPU: the MFMA
pipes or the memory system?  (GPU box; build tools/csrc/libprobe.so first:
 hipcc --offload-arch=gfx950 -O3 -shared -fPIC -x hip tools/csrc/probe_kernels.hip -o tools/csrc/libprobe.so)

Times danet_lstm_bwd / danet_lstm_fwd alone, next to a pure-MFMA burner and next to a pure
memory streamer (256 persistent workgroups each, on a side stream).'''
import ctypes
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
P = ctypes.CDLL(os.path.join(ROOT, 'tools', 'csrc', 'libprobe.so'))
P.probe_mfma.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
P.probe_mfma_duty.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
P.probe_mem.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                        ctypes.c_long, ctypes.c_int]


def main():
    B, T, H = 32, 128, 300
    dev = torch.device('cuda')
    torch.manual_seed(0)
    gx = [torch.randn(T * B, 4 * H, device=dev) * 0.5 for _ in range(2)]
    Wh = [torch.randn(H, 4 * H, device=dev) * (0.75 / H ** 0.5) for _ in range(2)]
    dy = torch.randn(T, B, 2 * H, device=dev)
    n = L.danet_lstm_workspace_bytes(T, B, H, 2)
    ws = torch.zeros(n, dtype=torch.uint8, device=dev)
    main_s = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    big = torch.randn(64 << 20, device=dev)          # 256 MB: streams from HBM
    small = torch.randn(1 << 20, device=dev)         # 4 MB: L2-resident
    sink = torch.zeros(16, device=dev)
    ypad = torch.empty(T + 2, B, 2 * H, device=dev)
    cells = [torch.empty(T * B, H, device=dev) for _ in range(2)]
    das = [torch.empty(T * B, 4 * H, device=dev) for _ in range(2)]
    gates = [x.clone() for x in gx]

    def fwd():
        for d in range(2):
            gates[d].copy_(gx[d])
        torch.cuda.synchronize()
        return lambda: _lib.check(L.danet_lstm_fwd(
            main_s.cuda_stream, T, B, H, 2, ptr(gates[0]), ptr(gates[1]), ptr(Wh[0]), ptr(Wh[1]),
            4 * H, ptr(ypad), 2 * H, ptr(gates[0]), ptr(gates[1]), ptr(cells[0]), ptr(cells[1]),
            ptr(ws), n, None, 0))

    def bwd():
        return lambda: _lib.check(L.danet_lstm_bwd(
            main_s.cuda_stream, T, B, H, 2, ptr(dy), 2 * H, ptr(Wh[0]), ptr(Wh[1]), 4 * H,
            ptr(gates[0]), ptr(gates[1]), ptr(cells[0]), ptr(cells[1]), ptr(das[0]), ptr(das[1]),
            ptr(ws), n, None))

    loads = {
        'alone': None,
        'mfma burner x256': lambda: P.probe_mfma(side.cuda_stream, 256, ptr(sink), 60000),
        'mfma burner x512': lambda: P.probe_mfma(side.cuda_stream, 512, ptr(sink), 30000),
        # same MFMA duty (~60 %: 8 x 64 or 16 x 32 clocks of MFMA, then 5 x 64 clocks asleep), two granularities
        'duty 60% 32x32x2 x256': lambda: P.probe_mfma_duty(side.cuda_stream, 256, ptr(sink), 0, 1500, 8, 5),
        'duty 60% 16x16x4 x256': lambda: P.probe_mfma_duty(side.cuda_stream, 256, ptr(sink), 1, 1500, 16, 5),
        'duty 60% 32x32x2 fine  ': lambda: P.probe_mfma_duty(side.cuda_stream, 256, ptr(sink), 0, 6000, 2, 1),
        'duty 60% 16x16x4 fine  ': lambda: P.probe_mfma_duty(side.cuda_stream, 256, ptr(sink), 1, 6000, 4, 1),
        'HBM streamer x256': lambda: P.probe_mem(side.cuda_stream, 256, ptr(big), ptr(sink), big.numel() // 4, 12),
        'L2 streamer x256': lambda: P.probe_mem(side.cuda_stream, 256, ptr(small), ptr(sink), small.numel() // 4, 700),
    }
    for kname, mk in (('lstm_fwd', fwd), ('lstm_bwd', bwd)):
        for lname, load in loads.items():
            ts = []
            for it in range(5):
                k = mk()
                torch.cuda.synchronize()
                if load is not None:
                    load()
                    e = torch.cuda.Event(); e.record(side)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ls0, ls1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(main_s)
                k()
                e1.record(main_s)
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            print('%-9s %-20s %.1f us  (%.2f us/step)' % (kname, lname, ts[len(ts) // 2], ts[len(ts) // 2] / T),
                  flush=True)
        fwd()()        # leave valid gates/cells for the BPTT runs
        torch.cuda.synchronize()


if __name__ == '__main__':
    main()
