// Process-wide tuning / diagnostic options of libdanet_hip.so (set through the C ABI:
// danet_set_option / danet_get_option, include/danet_hip.h).  The library NEVER reads the
// process environment: a host that wants environment overrides applies them itself through
// the setter (the Python layer does, danet-tensorflow_amd/_lib.py).  Values are ints held in
// relaxed atomics, so launches from several host threads read a consistent value each.
#pragma once

enum DanetOpt {
  OPT_GEMM_DMA = 0,          // bit mask: 1 tile launches, 2 stream-K / grouped alone, 4 capped groups
  OPT_SPLITK_TARGET,         // workgroups a split-K product aims for
  OPT_GEMM_WGS,              // persistent workgroups of a stream-K launch (multiple of 8, <= 1024)
  OPT_GEMM_MAXSPLIT,         // workgroups that may share one stream-K tile
  OPT_GEMM_YIELD,            // capped groups sleep n*64 clocks per k-tile
  OPT_LSTM_FWD_UN,           // 0 auto | 8 | 12 units per forward workgroup
  OPT_LSTM_BWD_S,            // 0 auto | twins per BPTT group
  OPT_LSTM_BWD_U,            // 0 auto | 8 | 16 | 32 units per BPTT producer group
  OPT_LSTM_SPIN_LIMIT,       // 0 default bound | >= 256 polls
  OPT_LSTM_FAULT_INJECT,     // 1: workgroup 0 of every launch exits without publishing (tests)
  OPT_LSTM_XMAP,             // -1 per-kernel default | 0 | 1 XCD-aware workgroup order
  OPT_LSTM_FWD_SMALL,        // 1 GEMV kernel for B <= 4 | 0 MFMA kernel
  OPT_LSTM_FWD_FUSED,        // -1 auto (B >= 24) | 0 off | 1 whenever supported
  OPT_LSTM_BWD_TWIN_XCD,     // 1 (default): the twins of a BPTT group share an XCD (L2 hits on their re-reads) | 0
  OPT_GEMM_MFMA16,           // 1: capped groups beside a recurrent kernel use the 16x16x4 k-loop | 0: 32x32x2
  OPT_GEMM_X6_PLAN,         // 0 modelled | MI + 16 * slices: rows per workgroup tile (64 MI, MI = 2..4) / K slices of danet_gemm_x6
  OPT_COUNT
};

int danet_opt(int id);
