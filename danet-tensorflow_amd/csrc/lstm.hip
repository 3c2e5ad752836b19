// Persistent recurrent LSTM kernels for gfx950: the sequential half of
// Model.lyr_lstm / _lyr_bilstm (reference main.py:76-132, app/modules.py:120-137,
// cell math app/ops.py:139-147: a=[x,h]W+b, g linear, i/f/o sigmoid,
// c'=i*g+f*c, h'=o*tanh(c')).  The input half (x_t*Wx+b for all t) is hoisted
// into one MFMA GEMM by the host; these kernels run all T dependent steps of
// both directions in ONE launch.
//
// Decomposition.  The recurrent matmul h[B,H] x Wh[H,4H] is latency-bound
// (T dependent steps), so Wh must not move: workgroup (dir, g, p) keeps the
// 4 gate columns of its 8 hidden units stationary in LDS for the whole
// sequence ([K/4][32][4] layout -> one conflict-free ds_read_b128 feeds four
// v_mfma_f32_16x16x4_f32), holds the cell state of its (batch row, unit)
// pairs in registers, and per step only all-gathers h_{t-1} from the other
// workgroups of its cluster through the layer's own output buffer.  Batch rows
// are independent recurrences, so they are split into clusters (dir, g) of
// 16*MT rows that never synchronise with each other.  A fragments come
// straight from global memory with 16-B sc1 (L1-bypassing) loads: lane (r,q)
// loads h[r][k0+4q..k0+4q+3] and the j-th MFMA of the group pairs element j
// with weight row k0+4q+j -- a permutation of the contraction index that both
// operands share, so no LDS staging or shuffles are needed.  K is split over
// the 4 waves (one per SIMD); partial tiles are reduced through LDS.  BPTT is
// the mirror image: stationary Wh^T slice, all-gather of da_{t+-1}.
//
// Inter-workgroup hand-off: THE PAYLOAD IS THE FLAG.  Every exchanged word
// (h_t[b][j] / da_t[b][n]) is written exactly ONCE per launch, to its own
// address.  The host call pre-fills the exchange buffer with the bit pattern
// 0xFFFFFFFF (a NaN no arithmetic here produces: hardware NaNs are canonical
// 0x7FC00000), producers publish with 4-byte write-through (sc1) stores, and
// consumers simply re-issue their sc1 fragment loads until no word equals the
// sentinel.  Each 4-byte word validates itself, so nothing depends on store
// ordering, dispatch order or XCD placement (guide G16, form R2 "the data IS
// the flag"), and the step pays one store->load latency instead of
// drain + barrier + flag + poll + load.  Every spin is bounded; a timeout sets
// the status word (ws word 0) instead of hanging.
#include "common.h"
#include "options.h"
#include <hip/hip_ext.h>
#include <atomic>
#include <type_traits>
#include <stdlib.h>

// Events armed with danet_next_launch_events ride on the recurrent kernel's own dispatch packet
// (start / stop time stamps of exactly that kernel: bench.py times the kernels this way, a
// hipEventRecord in front of the launch would let the side stream's work overtake it).  Every
// entry point takes them first thing, so a rejected call leaves nothing armed.
void dn_take_launch_events(hipEvent_t* start, hipEvent_t* stop);   // gemm_f32.hip
#define DN_LAUNCH_EV(K_, GRID_, BLOCK_, LDS_, STREAM_, ARGS_)                                        \
  do {                                                                                               \
    if (ev_start || ev_stop)                                                                         \
      hipExtLaunchKernelGGL(K_, dim3(GRID_), dim3(BLOCK_), LDS_, STREAM_, ev_start, ev_stop, 0, ARGS_); \
    else K_<<<GRID_, BLOCK_, LDS_, STREAM_>>>(ARGS_);                                                \
  } while (0)

#define LSTM_UNITS_FWD 8    // hidden units per workgroup (x4 gates = 32 columns)
#define SPIN_LIMIT (1u << 20)
#define FWD_CH 5            // k-groups (16 k each) per wave whose loads fly together
#define SENTINEL 0xFFFFFFFFu

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// NB: the b128 buffer-load builtin must be assigned to a GCC-style
// __vector_size__ vector; assigning it to an ext_vector_type silently lowers to
// a single-dword load (hipcc 7.2).
typedef unsigned v4u __attribute__((__vector_size__(16)));

// 16-B L1-bypassing (sc1) load through a buffer descriptor; out-of-range
// offsets return 0 (which is != SENTINEL, i.e. "valid").
__device__ __forceinline__ v4u load_sc1_b128(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16 /*sc1*/);
}
// ... with the part of the offset that is the same for the whole wave (the time block) in the
// instruction's SCALAR offset: no vector instruction per load and step
__device__ __forceinline__ v4u load_sc1_b128s(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 16 /*sc1*/);
}
__device__ __forceinline__ bool has_sentinel(v4u v) {
  return v[0] == SENTINEL || v[1] == SENTINEL || v[2] == SENTINEL || v[3] == SENTINEL;
}
// the sentinel is the LARGEST 32-bit pattern: "any word of these is unpublished" == "their unsigned
// maximum is the sentinel" -- a tree of v_max3_u32 and one compare instead of a compare and a scalar
// AND per word (the validation of five 16-byte fragments was ~40 dependent instructions per step)
__device__ __forceinline__ unsigned umax4(v4u v) { return max(max(v[0], v[1]), max(v[2], v[3])); }

struct LstmFwdArgs {
  const float* gx[2];
  const float* Wh[2];
  float* gates[2];
  float* cell[2];
  float* ypad;
  int* status;
  int T, B, H, ndir, ldy, ldw, P, G, KP;  // KP = H padded to 16
  int xmap;   // consecutive block ids cycle over the clusters (placement, see kernel)
  unsigned spin_limit;   // bound of every inter-workgroup wait (validation passes)
  int fault;             // test hook: workgroup 0 exits without publishing (forces a timeout)
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// one failed validation pass: back off, and give up (once, for everybody) when
// the bound is hit.  Returns true when the caller must stop waiting.
__device__ __forceinline__ bool spin_fail(unsigned& spins, int* status, int lane, unsigned limit) {
  // the fragment loads are ordinary (non-volatile) reads to the compiler: this
  // clobber is what forces them to be re-issued on the next pass
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_sleep(2);
  ++spins;
  if ((spins & 255u) == 0) {
    if (__hip_atomic_load(status, RLX_AGENT) != 0) return true;   // someone gave up
    if (spins >= limit) {
      if (lane == 0) __hip_atomic_store(status, (int)DANET_STATUS_TIMEOUT, RLX_AGENT);
      return true;
    }
  }
  return false;
}

// Optional diagnostic build (-DDANET_LSTM_TRACE, tools/trace_lstm.py): workgroup 0
// wave 0 stamps the phases of every step with the 100 MHz wall clock into
// ws[64 ...] as [T][8] u64 (A top, B exchange valid, C MFMA done, D barrier 1,
// E published, F barrier 2, G retry count).
#ifdef DANET_LSTM_TRACE
#define TRACE(slot) do { if (blockIdx.x == 0 && threadIdx.x == 0) \
    ((unsigned long long*)((char*)a.status + 64))[(size_t)s * 8 + (slot)] = wall_clock64(); } while (0)
#define TRACE_VAL(slot, v) do { if (blockIdx.x == 0 && threadIdx.x == 0) \
    ((unsigned long long*)((char*)a.status + 64))[(size_t)s * 8 + (slot)] = (v); } while (0)
#define TRACE_AT(thr, slot) do { if (blockIdx.x == 0 && threadIdx.x == (thr)) \
    ((unsigned long long*)((char*)a.status_ws + 64))[(size_t)s * 8 + (slot)] = wall_clock64(); } while (0)
#define TRACE_AT_VAL(thr, slot, v) do { if (blockIdx.x == 0 && threadIdx.x == (thr)) \
    ((unsigned long long*)((char*)a.status_ws + 64))[(size_t)s * 8 + (slot)] = (v); } while (0)
#define TRACE_BYTES(T) ((size_t)(T) * 64)
#else
#define TRACE_AT(thr, slot) do {} while (0)
#define TRACE_AT_VAL(thr, slot, v) do {} while (0)
#define TRACE(slot) do {} while (0)
#define TRACE_VAL(slot, v) do {} while (0)
#define TRACE_BYTES(T) ((size_t)0)
#endif

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
// NW waves per workgroup split K (4 = one per SIMD; 8 / 16 shorten each wave's
// chain of sc1 exchange loads, which is what bounds the exchange phase).
// UN hidden units per workgroup (x 4 gates = NCOL columns = NT MFMA column tiles).  UN = 8 is the
// default; UN = 12 serves wide layers (H = 600): 50 instead of 75 workgroups per cluster make
// 16-row clusters fit the GPU (200 workgroups), which halves the h_{t-1} payload every workgroup
// reads per step (38 instead of 77 KB) and its MFMA count (120 instead of 160 per wave).
// Round 5: the product runs on the bf16 matrix cores like the fused kernel's recurrent half (six bf16
// piece products per fp32 product, split_pair in csrc/common.h, the exchange untouched): a lane's two
// 4-float fragments of k-groups 2b and 2b+1 are the 8 k of its A operand in k-block b.  At H = 600 the
// fp32 instructions were 120 x 32 clocks per wave and step of a 4.6 us step; 90 x 17 + the split now.
// k-groups are taken in chunks of FWD_CHP = 6 (three k-blocks) whose loads fly together; the weight
// pieces of a wave's first chunk are stationary in registers, those of later chunks (H > 384) in LDS,
// already in operand order ([wave][block][piece][column tile][lane] x 16 bytes).
#define FWD_CHP 6
template <int MT, int NW, int UN>
__global__ __launch_bounds__(64 * NW) void lstm_fwd_kernel(LstmFwdArgs a) {
  constexpr int CH = FWD_CHP, NBC = CH / 2;          // k-groups / k-blocks per chunk
  constexpr int NCOL = 4 * UN, NT = NCOL / 16;
  static_assert(NCOL % 16 == 0 && 16 * MT * UN <= 64 * NW, "ownership map: one thread per (row, unit)");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // smem: red [NW waves][16*MT][NCOL+1] | weight pieces of the chunks behind the first
  float* red = smem;
  u32x4v* Wp = reinterpret_cast<u32x4v*>(smem + ((NW * 16 * MT * (NCOL + 1) + 3) & ~3));

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bid = blockIdx.x;
  if (a.fault && bid == 0) return;   // test hook (DANET_LSTM_FAULT_INJECT): never publishes
  const int ncl = a.ndir * a.G;
  const int cl = a.xmap ? bid % ncl : bid / a.P;   // xmap: members of a cluster share bid % 8
  const int dir = cl / a.G, grp = cl % a.G;
  const int p = a.xmap ? bid / ncl : bid % a.P;
  const int H = a.H, B = a.B, T = a.T;
  const int u0 = p * UN;
  const int b0 = grp * (16 * MT);

  const unsigned ybytes = (unsigned)((size_t)(T + 2) * B * a.ldy * sizeof(float));
  const __amdgpu_buffer_rsrc_t yres = make_rsrc(a.ypad, ybytes);

  // gate-math ownership: thread -> (batch row bl, unit u)
  const int bl = tid / UN, ul = tid % UN;
  const bool owner = (bl < 16 * MT) && (b0 + bl < B) && (u0 + ul < H);
  const int bg = b0 + bl, unit = u0 + ul;
  float c_state = 0.f;

  // A-fragment addressing: lane (r = lane&15, q = lane>>4)
  const int fr = lane & 15, fq = lane >> 4;
  const int NG = a.KP / 16;
  const int NGW = (NG + NW - 1) / NW;                 // k-groups per wave
  const int NBW = (NGW + 1) / 2;                      // k-blocks per wave

  // B fragment of (wave, k-block b, column tile nt) for lane (n = fr, kq = fq): elements 0..3 =
  // k-group 2b of this wave, 4..7 = k-group 2b+1, k = 16 (g * NW + wave) + 4 kq + j; three pieces
  auto wfrag = [&](int b, int nt, uint32_t (&o)[3][4]) {
    const float* W = a.Wh[dir];
    const int n = nt * 16 + fr;
    const int gate = n / UN, u = u0 + (n % UN);
    float w[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = ((2 * b + (e >> 2)) * NW + wave) * 16 + fq * 4 + (e & 3);
      w[e] = (k < H && u < H) ? W[(size_t)k * a.ldw + gate * H + u] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) split_pair(w[2 * q], w[2 * q + 1], o[0][q], o[1][q], o[2][q]);
  };
  uint32_t wreg[NBC][NT][3][4];
#pragma unroll
  for (int b = 0; b < NBC; ++b)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wfrag(b, nt, wreg[b][nt]);
  for (int b = NBC; b < NBW; ++b)
    for (int nt = 0; nt < NT; ++nt) {
      uint32_t o[3][4];
      wfrag(b, nt, o);
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
        Wp[(((size_t)wave * (NBW - NBC) + (b - NBC)) * 3 + pc) * NT * 64 + nt * 64 + lane] =
            (u32x4v){o[pc][0], o[pc][1], o[pc][2], o[pc][3]};
    }
  __syncthreads();

  // owner threads: where step s publishes h / saves its activations; the pointers walk the time axis
  const int t_first_w = dir ? (T - 1) : 0;
  const ptrdiff_t tdir_w = dir ? -1 : 1;
  const ptrdiff_t hstep_w = tdir_w * (ptrdiff_t)B * a.ldy, gstep_w = tdir_w * (ptrdiff_t)B * (4 * H),
                  cstep_w = tdir_w * (ptrdiff_t)B * H;
  float* hp_w = a.ypad + (owner ? ((size_t)(t_first_w + 1) * B + bg) * a.ldy + dir * H + unit : 0);
  float* gs_w = a.gates[dir] + (owner ? ((size_t)t_first_w * B + bg) * (4 * H) + unit : 0);
  float* cs_w = a.cell[dir] + (owner ? ((size_t)t_first_w * B + bg) * H + unit : 0);
#define FW_OP(P_) __builtin_bit_cast(bf16x8v, (u32x4v){(P_)[0], (P_)[1], (P_)[2], (P_)[3]})
  for (int s = 0; s < T; ++s) {
    const int t = dir ? (T - 1 - s) : s;
    const int blk_prev = dir ? (t + 2) : t;  // ypad block holding h_{prev}

    TRACE(0);
    // prefetch this step's hoisted input projections (independent of h)
    float gxv[4] = {0.f, 0.f, 0.f, 0.f};
    if (owner) {
      const float* gp = a.gx[dir] + ((size_t)t * B + bg) * (4 * H) + unit;
#pragma unroll
      for (int gte = 0; gte < 4; ++gte) gxv[gte] = gp[gte * H];
    }

    f32x4 acc[MT][NT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt][0] = acc[mt][nt][1] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // step 0 multiplies the zero initial state (main.py:108-123): skip it.
    // This wave's k-groups are wave, wave+4, ...; all of a chunk's 16-B loads
    // are issued together and re-issued until every word has been published.
    if (s > 0) {
      for (int g0 = 0; g0 < NGW; g0 += CH) {
        v4u av[CH][MT];
        unsigned spins = 0;
        for (;;) {
          bool ok = true;
#pragma unroll
          for (int g = 0; g < CH; ++g) {
            const int k = ((g0 + g) * NW + wave) * 16 + fq * 4;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              const int row = b0 + mt * 16 + fr;
              unsigned off = ybytes;  // == num_records: out of range -> 0 (valid)
              if (row < B && k < H)
                off = (unsigned)((((size_t)blk_prev * B + row) * a.ldy + dir * H + k) * 4);
              av[g][mt] = load_sc1_b128(yres, off);
            }
          }
          unsigned mx = 0u;
#pragma unroll
          for (int g = 0; g < CH; ++g)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) mx = max(mx, umax4(av[g][mt]));
          ok = (mx != SENTINEL);
          if (__all(ok)) break;
          if (spin_fail(spins, a.status, lane, a.spin_limit)) break;
        }
        TRACE(1); TRACE_VAL(6, spins);
        // NO per-block branch below: a branch splits the accumulator chain into basic blocks
        // (measured 2x on the matrix phase with the fp32 instructions).  Blocks past the wave's
        // last multiply zeros (their loads were out of range) by finite weights.
#pragma unroll
        for (int b = 0; b < NBC; ++b) {
          uint32_t wq[NT][3][4];
          if (g0 == 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
              for (int pc = 0; pc < 3; ++pc)
#pragma unroll
                for (int q = 0; q < 4; ++q) wq[nt][pc][q] = wreg[b][nt][pc][q];
          } else {
            const int bb = min(g0 / 2 + b, NBW - 1) - NBC;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
              for (int pc = 0; pc < 3; ++pc) {
                const u32x4v v = Wp[(((size_t)wave * (NBW - NBC) + bb) * 3 + pc) * NT * 64 + nt * 64 + lane];
                wq[nt][pc][0] = v[0]; wq[nt][pc][1] = v[1]; wq[nt][pc][2] = v[2]; wq[nt][pc][3] = v[3];
              }
          }
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            // NB: bit_cast the WHOLE vector (hipcc 7.2 reads element 0 for every element otherwise)
            const f32x4 lo = __builtin_bit_cast(f32x4, av[2 * b][mt]);
            const f32x4 hi = __builtin_bit_cast(f32x4, av[2 * b + 1][mt]);
            uint32_t ap[3][4];
            split_pair(lo[0], lo[1], ap[0][0], ap[1][0], ap[2][0]);
            split_pair(lo[2], lo[3], ap[0][1], ap[1][1], ap[2][1]);
            split_pair(hi[0], hi[1], ap[0][2], ap[1][2], ap[2][2]);
            split_pair(hi[2], hi[3], ap[0][3], ap[1][3], ap[2][3]);
            // six piece products per column tile, small terms first (A piece, B piece)
#pragma unroll
            for (int t6 = 0; t6 < 6; ++t6) {
              constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
              const bf16x8v af = FW_OP(ap[PA[t6]]);
#pragma unroll
              for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt][t6 & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, FW_OP(wq[nt][PB[t6]]),
                                                                              acc[mt][nt][t6 & 1], 0, 0, 0);
            }
          }
        }
      }
    }


    TRACE(2);
    // cross-wave reduction.  D layout 16x16: col = lane&15, row = 4*(lane>>4)+r
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          red[(wave * 16 * MT + mt * 16 + 4 * fq + r) * (NCOL + 1) + nt * 16 + fr] = acc[mt][nt][0][r] + acc[mt][nt][1][r];
    __syncthreads();
    TRACE(3);

    if (owner) {
      float pre[4];
#pragma unroll
      for (int gte = 0; gte < 4; ++gte) {
        float v = gxv[gte];
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[(w * 16 * MT + bl) * (NCOL + 1) + gte * UN + ul];
        pre[gte] = v;
      }
      const float g = pre[0];                 // linear candidate (ops.py:143)
      const float ig = sigmoid_hw(pre[1]);
      const float fg = sigmoid_hw(pre[2]);
      const float og = sigmoid_hw(pre[3]);
      c_state = ig * g + fg * c_state;        // ops.py:146
      const float h = og * tanh_hw(c_state);   // ops.py:147
      // publish h_t: one write-through 4-byte store, no drain, no flag (owner pointers walk the
      // time axis: no 64-bit multiplies in front of it)
      __hip_atomic_store(hp_w, h, RLX_AGENT);
      // saved activations are only read by later kernels (plain stores)
      gs_w[0] = g; gs_w[H] = ig; gs_w[2 * H] = fg; gs_w[3 * H] = og;
      cs_w[0] = c_state;
      hp_w += hstep_w; gs_w += gstep_w; cs_w += cstep_w;
    }
    // `red` is rewritten only after this wave has seen every h_t word of its
    // cluster, i.e. after every owner thread (in every wave) has finished
    // reading `red` for this step -- but waves without owners never publish,
    // so a block barrier keeps the reuse safe in all shapes.
    TRACE(4);
    __syncthreads();
    TRACE(5);
  }
}

// ---------------------------------------------------------------------------
// forward, tiny batch (B <= 4): the demo / inference path (main.py:623-627, B = 1)
// ---------------------------------------------------------------------------
// With one batch row a 16-row MFMA tile is 1/16 used and the 40 MFMAs per wave and step
// (0.64 us) are the largest phase of the 1.83 us step.  Same decomposition, hand-off and
// weight layout as lstm_fwd_kernel, but the product is a GEMV on the vector ALU: lane
// (fr, fq) holds the weights of columns fr and 16+fr for k = 16 kg + 4 fq + j (the MFMA
// fragment layout, reused), loads h[row][k..k+3] for each of the <= 4 rows, does 8 FMAs
// per row and k-group, and the four fq lane groups are summed with two DPP-style shuffles
// before the usual cross-wave reduction through LDS.
template <int RMAX>
__global__ __launch_bounds__(256) void lstm_fwd_small_kernel(LstmFwdArgs a) {
  constexpr int NW = 4, CH = FWD_CH;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Wl = smem;
  float* red = smem + (size_t)a.KP * 32;     // [NW][RMAX][33]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bid = blockIdx.x;
  if (a.fault && bid == 0) return;
  const int dir = a.xmap ? bid % a.ndir : bid / a.P;
  const int p = a.xmap ? bid / a.ndir : bid % a.P;
  const int H = a.H, B = a.B, T = a.T;
  const int u0 = p * LSTM_UNITS_FWD;

  {
    const float* W = a.Wh[dir];
    for (int idx = tid; idx < a.KP * 32; idx += 64 * NW) {
      const int k = idx >> 5, n = idx & 31;
      const int gate = n >> 3, u = u0 + (n & 7);
      float v = 0.f;
      if (k < H && u < H) v = W[(size_t)k * a.ldw + gate * H + u];
      Wl[((k >> 2) * 32 + n) * 4 + (k & 3)] = v;
    }
  }
  __syncthreads();

  const unsigned ybytes = (unsigned)((size_t)(T + 2) * B * a.ldy * sizeof(float));
  const __amdgpu_buffer_rsrc_t yres = make_rsrc(a.ypad, ybytes);

  // gate-math ownership: thread -> (batch row bl, unit ul), bl < B <= RMAX
  const int bl = tid >> 3, ul = tid & 7;
  const bool owner = (bl < B) && (u0 + ul < H);
  const int unit = u0 + ul;
  float c_state = 0.f;

  const int fr = lane & 15, fq = lane >> 4;
  const int NG = a.KP / 16;
  f32x4 wreg[CH][2];
#pragma unroll
  for (int g = 0; g < CH; ++g) {
    const int kg = g * NW + wave;
    const int k4 = (kg < NG ? kg : 0) * 4 + fq;
    wreg[g][0] = *reinterpret_cast<const f32x4*>(&Wl[(k4 * 32 + fr) * 4]);
    wreg[g][1] = *reinterpret_cast<const f32x4*>(&Wl[(k4 * 32 + 16 + fr) * 4]);
    if (kg >= NG) { wreg[g][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; wreg[g][1] = wreg[g][0]; }
  }

  for (int s = 0; s < T; ++s) {
    const int t = dir ? (T - 1 - s) : s;
    const int blk_prev = dir ? (t + 2) : t;
    float gxv[4] = {0.f, 0.f, 0.f, 0.f};
    if (owner) {
      const float* gp = a.gx[dir] + ((size_t)t * B + bl) * (4 * H) + unit;
#pragma unroll
      for (int gte = 0; gte < 4; ++gte) gxv[gte] = gp[gte * H];
    }
    float acc[RMAX][2];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) acc[r][0] = acc[r][1] = 0.f;

    if (s > 0) {
      v4u av[CH][RMAX];
      unsigned spins = 0;
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int g = 0; g < CH; ++g) {
          const int k = (g * NW + wave) * 16 + fq * 4;
#pragma unroll
          for (int r = 0; r < RMAX; ++r) {
            unsigned off = ybytes;     // out of range -> 0 (valid)
            if (r < B && k < H)
              off = (unsigned)((((size_t)blk_prev * B + r) * a.ldy + dir * H + k) * 4);
            av[g][r] = load_sc1_b128(yres, off);
          }
        }
#pragma unroll
        for (int g = 0; g < CH; ++g)
#pragma unroll
          for (int r = 0; r < RMAX; ++r) ok &= !has_sentinel(av[g][r]);
        if (__all(ok)) break;
        if (spin_fail(spins, a.status, lane, a.spin_limit)) break;
      }
#pragma unroll
      for (int g = 0; g < CH; ++g) {
        const f32x4 w0 = wreg[g][0], w1 = wreg[g][1];
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
          const f32x4 hf = __builtin_bit_cast(f32x4, av[g][r]);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[r][0] = fmaf(hf[j], w0[j], acc[r][0]);
            acc[r][1] = fmaf(hf[j], w1[j], acc[r][1]);
          }
        }
      }
    }
    // sum the four fq lane groups (lanes fr, fr+16, fr+32, fr+48), then the waves via LDS
#pragma unroll
    for (int r = 0; r < RMAX; ++r)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        float v = acc[r][nt];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        acc[r][nt] = v;
      }
    if (fq == 0) {
#pragma unroll
      for (int r = 0; r < RMAX; ++r) {
        red[(wave * RMAX + r) * 33 + fr] = acc[r][0];
        red[(wave * RMAX + r) * 33 + 16 + fr] = acc[r][1];
      }
    }
    __syncthreads();

    if (owner) {
      float pre[4];
#pragma unroll
      for (int gte = 0; gte < 4; ++gte) {
        float v = gxv[gte];
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[(w * RMAX + bl) * 33 + gte * 8 + ul];
        pre[gte] = v;
      }
      const float g = pre[0];
      const float ig = sigmoid_hw(pre[1]);
      const float fg = sigmoid_hw(pre[2]);
      const float og = sigmoid_hw(pre[3]);
      c_state = ig * g + fg * c_state;
      const float h = og * tanh_hw(c_state);
      float* hp = a.ypad + ((size_t)(t + 1) * B + bl) * a.ldy + dir * H + unit;
      __hip_atomic_store(hp, h, RLX_AGENT);
      float* gs = a.gates[dir] + ((size_t)t * B + bl) * (4 * H) + unit;
      gs[0] = g; gs[H] = ig; gs[2 * H] = fg; gs[3 * H] = og;
      a.cell[dir][((size_t)t * B + bl) * H + unit] = c_state;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// forward, input projection fused (the whole cell of app/ops.py:139-147 in one kernel)
// ---------------------------------------------------------------------------
// The exchange of h_{t-1} costs ~1 us per step during which the matrix cores of the
// recurrent workgroups idle (152 of 256 CUs at cfg 2, one MFMA chain of 0.66 us per
// 2.3 us step).  The input half x_t*Wx of a = [x_t, h_{t-1}] W + b does not depend on the
// recurrence, so this kernel computes it IN that window instead of in a hoisted GEMM:
//   - the workgroup's 32 columns of Wx (all D rows) are stationary in REGISTERS, K split
//     over the 4 waves like Wh (D <= 640: <= 10 k-groups of 16 per wave, 80 VGPRs);
//   - per step a wave runs a quarter of the x_t*Wx MFMAs, issues the sc1 exchange loads of
//     h_{t-1} (not earlier: a first-attempt load that races a slower producer is only
//     re-issued after the MFMA block), issues the prefetch of x_{t+1}, runs the rest of the
//     MFMAs while the loads fly, validates, and continues the SAME accumulators with
//     h_{t-1}*Wh; reduction, gate math and publish are those of lstm_fwd_kernel (bias added
//     in the gate phase).
// Two things the compiler does with such a loop had to be defeated (each cost ~1 us per
// step): (1) a loop-carried prefetch is waited for at the loop latch with s_waitcnt vmcnt(0),
// which on gfx9 also waits for the write-through publish stores issued just before -- the
// prefetched registers are therefore consumed BEFORE the publish; (2) MFMAs have no memory
// effect and are moved across barriers / waits unless pinned with sched_barrier.
// (A wave-specialised variant -- separate exchange and matrix waves -- was slower: MFMA
// issue from one wave starves the other waves of its SIMD, so the gate math of the exchange
// waves waited for the whole MFMA block; profiles/r02_b_fused_fwd_trace.txt.)
// Envelope: 16-row clusters (MT = 1), H <= 320 (every Wh fragment in registers), D <= 640;
// other shapes take the hoisted-GEMM path (danet_lstm_fwd).
struct LstmFwdFxArgs {
  const float* x;        // [T][B][ldx] time-major layer input, columns >= D finite (zero pad)
  const float* W[2];     // [D+H][ldw]: rows 0..D-1 input half, D..D+H-1 recurrent half
  const float* bias[2];  // [4H]
  float* gates[2];
  float* cell[2];
  float* ypad;
  int* status;
  void* status_ws;       // workspace base (trace records of the diagnostic build)
  int T, B, H, D, ndir, ldx, ldy, ldw, P, G, KP, DP;   // KP / DP = H / D padded to 16
  int xmap;
  unsigned spin_limit;
  int fault;
};

// Input-half k-groups (16 k each): CHX "window" groups per wave, computed in the exchange wait of
// the step they belong to, plus CHE "early" groups owned by waves 2 and 3 only, computed one step
// AHEAD while waves 0 and 1 do the gate math of the previous step (the 128 owner threads are all in
// waves 0 and 1; waves 2 and 3 used to idle through that phase).  k-group of (wave, window g) =
// g * 4 + wave; of (wave >= 2, early e) = 4 * CHX + 2 * e + wave - 2.
//
// Round 5: the RECURRENT half runs on the bf16 matrix cores.  h_{t-1} * Wh on v_mfma_f32_16x16x4_f32
// was 40 dependent-issue matrix instructions of 32 clocks per wave and step (0.58 us, all of it
// behind the exchange).  Both operands are now split EXACTLY into three bf16 pieces (csrc/common.h
// split_pair; six of the nine piece products accumulated in fp32, the dropped ones < 2^-25 |h||w|:
// the arithmetic of csrc/gemm_x6.hip) and one v_mfma_f32_16x16x32_bf16 covers 32 k in 17 clocks: 36
// instructions.  The exchange is untouched (fp32 words in the layer's output buffer, the same five
// 16-byte loads per lane): a lane's two 4-float fragments of k-groups 2b and 2b+1 ARE the 8 k of its
// A operand in k-block b -- the weights are laid out for that k set.  The INPUT half follows the same
// scheme (x stays fp32 in memory, the same loads; a window / early k-block = two of the wave's window /
// early k-groups): 66 instead of 88 matrix instructions per wave and step at D = 600, 17 instead of 32
// clocks each, behind 36 vector instructions of split per k-block -- 2.74 -> 2.53 us per time step.
// (Splitting at the PRODUCER instead, with
// the pieces as the exchange payload, was built first and lost 0.7 us per step to the hand-off:
// profiles/r05_b_fwd_producer_split.txt.)
template <int CHX, int CHE>
__global__ __launch_bounds__(256) void lstm_fwd_fx_kernel(LstmFwdFxArgs a) {
  constexpr int NW = 4, CH = FWD_CH;
  constexpr int NB = (CH + 1) / 2;      // k-blocks of 32 = pairs of 16-k groups per wave
  constexpr int XBA = 1;   // k-blocks of the input half done before the exchange loads are issued (0: retries)
  // ... and k-groups of it where it stays on the fp32 instructions (the bottom layer): ALL of them.
  // With the gate phase at 45 vector instructions (hardware reciprocal, walking pointers) the next
  // step's top comes 0.1 us sooner after the publish, and exchange loads issued before the
  // published words are visible do not fail, they come back LATE (publish -> valid 0.93 -> 1.15 us):
  // 2 groups in front of the loads 273 us per launch averaged over the three layers, 1 group 278,
  // none 305 (profiles/r05_k_fwd_load_point.txt).
  constexpr int XGA = CHX;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // smem: recurrent weights [KP/4][32][4] (read once into registers) | red [NW][16][33]
  float* Wl = smem;
  float* red = smem + (size_t)a.KP * 32;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool early = wave >= 2;      // uniform
  const int bid = blockIdx.x;
  if (a.fault && bid == 0) return;   // test hook (DANET_LSTM_FAULT_INJECT): never publishes
  const int ncl = a.ndir * a.G;
  const int cl = a.xmap ? bid % ncl : bid / a.P;
  const int dir = cl / a.G, grp = cl % a.G;
  const int p = a.xmap ? bid / ncl : bid % a.P;
  const int H = a.H, B = a.B, T = a.T, D = a.D;
  const int u0 = p * LSTM_UNITS_FWD;
  const int b0 = grp * 16;
  const float* Wd = a.W[dir];

  {
    const float* W = Wd + (size_t)D * a.ldw;     // recurrent rows
    for (int idx = tid; idx < a.KP * 32; idx += 64 * NW) {
      const int k = idx >> 5, n = idx & 31;
      const int gate = n >> 3, u = u0 + (n & 7);
      float v = 0.f;
      if (k < H && u < H) v = W[(size_t)k * a.ldw + gate * H + u];
      Wl[((k >> 2) * 32 + n) * 4 + (k & 3)] = v;
    }
  }
  __syncthreads();

  const unsigned ybytes = (unsigned)((size_t)(T + 2) * B * a.ldy * sizeof(float));
  const __amdgpu_buffer_rsrc_t yres = make_rsrc(a.ypad, ybytes);
  const unsigned xbytes = (unsigned)((size_t)T * B * a.ldx * sizeof(float));
  const __amdgpu_buffer_rsrc_t xres = make_rsrc(a.x, xbytes);

  const int bl = tid >> 3, ul = tid & 7;
  const bool owner = (bl < 16) && (b0 + bl < B) && (u0 + ul < H);
  const int bg = b0 + bl, unit = u0 + ul;
  float c_state = 0.f;
  float bq[4] = {0.f, 0.f, 0.f, 0.f};
  if (owner) {
#pragma unroll
    for (int gte = 0; gte < 4; ++gte) bq[gte] = a.bias[dir][gte * H + unit];
  }

  const int fr = lane & 15, fq = lane >> 4;
  const int NG = a.KP / 16;

  // stationary fragments: recurrent half (bf16 pieces, from LDS) and input half (straight from global).
  // B operand of v_mfma_f32_16x16x32_bf16: lane (n = lane % 16, kq = lane / 16) holds column n's
  // weights of the 8 k this lane GROUP's A fragments carry in k-block b: elements 0..3 = k-group 2b,
  // elements 4..7 = k-group 2b+1, k = 16 (g * NW + wave) + 4 kq + j each.
  uint32_t wreg[NB][3][2][4];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      float w[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int g = 2 * b + (e >> 2);
        const int k = (g * NW + wave) * 16 + fq * 4 + (e & 3);
        w[e] = (g < CH && k < a.KP) ? Wl[((k >> 2) * 32 + nt * 16 + fr) * 4 + (k & 3)] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        split_pair(w[2 * q], w[2 * q + 1], wreg[b][0][nt][q], wreg[b][1][nt][q], wreg[b][2][nt][q]);
    }
  // input half: the same six-piece arithmetic, k-block b = window k-groups 2b | 2b+1 of this wave
  // (early list: early k-groups 2b | 2b+1; an odd last group is paired with zeros)
  // XBF: at D <= 160 (CHX = 2, the bottom layer: 10 k-groups in all) the fp32 instructions stay -- one
  // k-block of split + 12 instructions in front of the exchange loads is longer than the 8 fp32
  // instructions it replaces, and that layer's step got 0.3 us slower with it (304 -> 324 us per launch)
  constexpr bool XBF = CHX >= 4;
  static_assert(CHX % 2 == 0, "window k-groups are paired into k-blocks");
  constexpr int XBW = XBF ? CHX / 2 : 1, XBE = XBF ? (CHE + 1) / 2 : 1;
  auto wxfrag = [&](int kga, int kgb, bool on, int nt, uint32_t (&o)[3][4]) {
    const int n = nt * 16 + fr;
    const int gate = n >> 3, u = u0 + (n & 7);
    float w[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int kg = (e < 4) ? kga : kgb;
      const int k = kg * 16 + fq * 4 + (e & 3);
      w[e] = (on && kg >= 0 && u < H && k < D) ? Wd[(size_t)k * a.ldw + gate * H + u] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) split_pair(w[2 * q], w[2 * q + 1], o[0][q], o[1][q], o[2][q]);
  };
  uint32_t wx[XBW][2][3][4];
#pragma unroll
  for (int b = 0; b < (XBF ? XBW : 0); ++b)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) wxfrag((2 * b) * NW + wave, (2 * b + 1) * NW + wave, true, nt, wx[b][nt]);
  uint32_t wxe[XBE][2][3][4];
#pragma unroll
  for (int b = 0; b < (XBF ? XBE : 0); ++b)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
      wxfrag(NW * CHX + 2 * (2 * b) + (wave - 2), (2 * b + 1 < CHE) ? NW * CHX + 2 * (2 * b + 1) + (wave - 2) : -1,
             early, nt, wxe[b][nt]);

  // (fp32 fragments for the narrow input of the bottom layer, see XBF)
  f32x4 wxf[XBF ? 1 : CHX][2];
#pragma unroll
  for (int g = 0; g < (XBF ? 0 : CHX); ++g) {
    const int k0 = (g * NW + wave) * 16 + fq * 4;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int n = nt * 16 + fr;
      const int gate = n >> 3, u = u0 + (n & 7);
      f32x4 w = {0.f, 0.f, 0.f, 0.f};
      if (u < H) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (k0 + j < D) w[j] = Wd[(size_t)(k0 + j) * a.ldw + gate * H + u];
      }
      wxf[g][nt] = w;
    }
  }

  f32x4 wxef[XBF ? 1 : CHE][2];
#pragma unroll
  for (int g = 0; g < (XBF ? 0 : CHE); ++g) {
    const int k0 = (NW * CHX + 2 * g + (wave - 2)) * 16 + fq * 4;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int n = nt * 16 + fr;
      const int gate = n >> 3, u = u0 + (n & 7);
      f32x4 w = {0.f, 0.f, 0.f, 0.f};
      if (early && u < H) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (k0 + j < D) w[j] = Wd[(size_t)(k0 + j) * a.ldw + gate * H + u];
      }
      wxef[g][nt] = w;
    }
  }

  // byte offsets of this lane's x / h fragments within one time block (out of range -> 0)
  unsigned xoff[CHX], xoffe[CHE], hcol[CH];
  {
    const int row = b0 + fr;
#pragma unroll
    for (int g = 0; g < CHX; ++g) {
      const int k0 = (g * NW + wave) * 16 + fq * 4;
      xoff[g] = (row < B && k0 < D) ? (unsigned)(((size_t)row * a.ldx + k0) * 4) : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int g = 0; g < CHE; ++g) {
      const int k0 = (NW * CHX + 2 * g + (wave - 2)) * 16 + fq * 4;
      xoffe[g] = (early && row < B && k0 < D) ? (unsigned)(((size_t)row * a.ldx + k0) * 4) : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int g = 0; g < CH; ++g) {
      const int k0 = (g * NW + wave) * 16 + fq * 4;
      hcol[g] = (row < B && k0 < H) ? (unsigned)(((size_t)row * a.ldy + dir * H + k0) * 4) : 0xFFFFFFFFu;
    }
  }
  const unsigned xblk = (unsigned)((size_t)B * a.ldx * 4);
  const unsigned yblk = (unsigned)((size_t)B * a.ldy * 4);
  // loop-invariant vector offsets (an offset >= the buffer's size reads zeros: the raw-buffer range check
  // covers the VECTOR offset, the scalar offset is outside it -- the standard shape depends on it at every
  // step: H = 300 leaves the lanes with k0 >= 300 of the 320-wide k range out of range, and a non-zero
  // fragment there would meet zero weights as 0 * garbage; tests/test_gpu_lstm.py covers H = 300 / 600 and
  // ragged B); the time block goes into the loads' scalar offset.  Step 0 reads h_{-1} from the zero pad block of ypad (block 0 /
  // T + 1), which no step ever writes.
  unsigned xoffv[CHX], xoffev[CHE], hcolv[CH];
#pragma unroll
  for (int g = 0; g < CHX; ++g) xoffv[g] = xoff[g] == 0xFFFFFFFFu ? xbytes : xoff[g];
#pragma unroll
  for (int g = 0; g < CHE; ++g) xoffev[g] = xoffe[g] == 0xFFFFFFFFu ? xbytes : xoffe[g];
#pragma unroll
  for (int g = 0; g < CH; ++g) hcolv[g] = hcol[g] == 0xFFFFFFFFu ? ybytes : hcol[g];
  v4u xc[CHX], xce[CHE];
  {
    const int t0 = dir ? (T - 1) : 0;
#pragma unroll
    for (int g = 0; g < CHX; ++g)
      xc[g] = __builtin_amdgcn_raw_buffer_load_b128(
          xres, xoff[g] == 0xFFFFFFFFu ? xbytes : xoff[g] + (unsigned)t0 * xblk, 0, 0);
#pragma unroll
    for (int g = 0; g < CHE; ++g)
      xce[g] = __builtin_amdgcn_raw_buffer_load_b128(
          xres, xoffe[g] == 0xFFFFFFFFu ? xbytes : xoffe[g] + (unsigned)t0 * xblk, 0, 0);
    // wait for them HERE: loads still pending at loop entry make the compiler put one
    // s_waitcnt vmcnt(0) into the loop header, where it then also waits for the previous
    // step's publish stores on every iteration
#pragma unroll
    for (int g = 0; g < CHX; ++g) asm volatile("" : "+v"(xc[g]));
#pragma unroll
    for (int g = 0; g < CHE; ++g) asm volatile("" : "+v"(xce[g]));
  }

#define FX_OPX(P_) __builtin_bit_cast(bf16x8v, (u32x4v){(P_)[0], (P_)[1], (P_)[2], (P_)[3]})
  // one k-block of the input half: split the lane's two x fragments, twelve matrix instructions
#define FX_BLOCKX(XLO_, XHI_, W_, ACC_)                                                        \
  do {                                                                                         \
    const f32x4 lo_ = __builtin_bit_cast(f32x4, XLO_), hi_ = __builtin_bit_cast(f32x4, XHI_);  \
    uint32_t xp_[3][4];                                                                        \
    split_pair(lo_[0], lo_[1], xp_[0][0], xp_[1][0], xp_[2][0]);                               \
    split_pair(lo_[2], lo_[3], xp_[0][1], xp_[1][1], xp_[2][1]);                               \
    split_pair(hi_[0], hi_[1], xp_[0][2], xp_[1][2], xp_[2][2]);                               \
    split_pair(hi_[2], hi_[3], xp_[0][3], xp_[1][3], xp_[2][3]);                               \
    _Pragma("unroll") for (int t6 = 0; t6 < 6; ++t6) {                                         \
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};                    \
      const bf16x8v af_ = FX_OPX(xp_[PA[t6]]);                                                 \
      ACC_[0][t6 & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af_, FX_OPX(W_[0][PB[t6]]), ACC_[0][t6 & 1], 0, 0, 0); \
      ACC_[1][t6 & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af_, FX_OPX(W_[1][PB[t6]]), ACC_[1][t6 & 1], 0, 0, 0); \
    }                                                                                          \
  } while (0)
#define FX_GROUPF(X_, W_, ACC_)                                                                \
  do {                                                                                         \
    const f32x4 xf_ = __builtin_bit_cast(f32x4, X_);                                           \
    const f32x4 w0_ = W_[0], w1_ = W_[1];                                                      \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                            \
      ACC_[0][j & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xf_[j], w0_[j], ACC_[0][j & 1], 0, 0, 0); \
      ACC_[1][j & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xf_[j], w1_[j], ACC_[1][j & 1], 0, 0, 0); \
    }                                                                                          \
  } while (0)
#define FX_GX(b) FX_BLOCKX(xc[2 * (b)], xc[2 * (b) + 1], wx[b], acc)
#define FX_GXE(b) FX_BLOCKX(xce[2 * (b)], ((2 * (b) + 1 < CHE) ? xce[(2 * (b) + 1 < CHE) ? 2 * (b) + 1 : 0] : (v4u){0u, 0u, 0u, 0u}), wxe[b], accn)

  // owner threads: where step s publishes h / saves its activations; the pointers walk the time axis
  const int t_first_w = dir ? (T - 1) : 0;
  const ptrdiff_t tdir_w = dir ? -1 : 1;
  const ptrdiff_t hstep_w = tdir_w * (ptrdiff_t)B * a.ldy, gstep_w = tdir_w * (ptrdiff_t)B * (4 * H),
                  cstep_w = tdir_w * (ptrdiff_t)B * H;
  float* hp_w = a.ypad + (owner ? ((size_t)(t_first_w + 1) * B + bg) * a.ldy + dir * H + unit : 0);
  float* gs_w = a.gates[dir] + (owner ? ((size_t)t_first_w * B + bg) * (4 * H) + unit : 0);
  float* cs_w = a.cell[dir] + (owner ? ((size_t)t_first_w * B + bg) * H + unit : 0);
  // early groups of step 0
  f32x4 accn[2][2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    accn[nt][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    accn[nt][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  if (early) {
    if constexpr (XBF) {
#pragma unroll
      for (int b = 0; b < XBE; ++b) FX_GXE(b);
    } else {
#pragma unroll
      for (int g = 0; g < CHE; ++g) FX_GROUPF(xce[g], wxef[g], accn);
    }
  }

  for (int s = 0; s < T; ++s) {
    const int t = dir ? (T - 1 - s) : s;
    const int blk_prev = dir ? (t + 2) : t;

    TRACE_AT(0, 0);
    f32x4 acc[2][2];       // starts from the early groups' sums (zeros in waves 0 and 1)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      acc[nt][0] = accn[nt][0];
      acc[nt][1] = accn[nt][1];
    }
    // (a) first quarter of the input half
    if constexpr (XBF) {
#pragma unroll
      for (int b = 0; b < XBA; ++b) FX_GX(b);
    } else {
#pragma unroll
      for (int g = 0; g < XGA; ++g) FX_GROUPF(xc[g], wxf[g], acc);
    }
    __builtin_amdgcn_sched_barrier(0);
    // (b) exchange loads of h_{t-1} (step 0: out of range -> zeros)
    v4u av[CH];
    const unsigned hso = (unsigned)blk_prev * yblk;
#pragma unroll
    for (int g = 0; g < CH; ++g) av[g] = load_sc1_b128s(yres, hcolv[g], hso);
    __builtin_amdgcn_sched_barrier(0);
    TRACE_AT(0, 1);
    // (c) rest of the input half while the loads fly
    if constexpr (XBF) {
#pragma unroll
      for (int b = XBA; b < XBW; ++b) FX_GX(b);
    } else {
#pragma unroll
      for (int g = XGA; g < CHX; ++g) FX_GROUPF(xc[g], wxf[g], acc);
    }
    __builtin_amdgcn_sched_barrier(0);
    TRACE_AT(0, 2);

    // (d) every exchanged word published?  re-issue until so (bounded)
    {
      unsigned spins = 0;
      for (;;) {
        unsigned mx = 0u;
#pragma unroll
        for (int g = 0; g < CH; ++g) mx = max(mx, umax4(av[g]));
        if (__all(mx != SENTINEL)) break;
        if (spin_fail(spins, a.status, lane, a.spin_limit)) break;
#pragma unroll
        for (int g = 0; g < CH; ++g) av[g] = load_sc1_b128s(yres, hcolv[g], hso);
      }
      TRACE_AT_VAL(0, 6, spins);
    }
    TRACE_AT(0, 3);
    // (e) recurrent half continues the same accumulators.  x of the next step is fetched HERE,
    // a few loads behind each k-group's MFMAs (a load costs ~60 clocks of issue, which then
    // overlap the matrix pipe's 256 clocks per group): at the top of the step the loads queued
    // behind the previous step's stores in the memory pipeline and delayed the exchange loads,
    // and as one block between validation and this phase they cost 0.2 us of the critical path.
    v4u xn[CHX], xne[CHE];
    const int tn = dir ? (t - 1) : (t + 1);
    const bool have = (s + 1 < T);
    const unsigned xso = have ? (unsigned)tn * xblk : 0u;     // (last step: any in-range block, the data is unused)
#pragma unroll
    for (int g = 0; g < CHE; ++g) xne[g] = (v4u){0u, 0u, 0u, 0u};
    __builtin_amdgcn_sched_barrier(0);
    // pieces of k-block b: [piece] x 4 words = the lane's 8 k as bf16 (groups 2b | 2b+1)
#define FX_SPLIT(B_, DST_)                                                                       \
    do {                                                                                         \
      const f32x4 lo_ = __builtin_bit_cast(f32x4, av[2 * (B_)]);                                 \
      const f32x4 hi_ = (2 * (B_) + 1 < CH) ? __builtin_bit_cast(f32x4, av[(2 * (B_) + 1 < CH) ? 2 * (B_) + 1 : 0]) \
                                            : (f32x4){0.f, 0.f, 0.f, 0.f};                       \
      split_pair(lo_[0], lo_[1], DST_[0][0], DST_[1][0], DST_[2][0]);                            \
      split_pair(lo_[2], lo_[3], DST_[0][1], DST_[1][1], DST_[2][1]);                            \
      split_pair(hi_[0], hi_[1], DST_[0][2], DST_[1][2], DST_[2][2]);                            \
      split_pair(hi_[2], hi_[3], DST_[0][3], DST_[1][3], DST_[2][3]);                            \
    } while (0)
    uint32_t ap[NB][3][4];
    FX_SPLIT(0, ap[0]);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (b > 0) FX_SPLIT(b, ap[b]);
      // six piece products per column tile, small terms first (A piece, B piece); the two column
      // tiles alternate, each on two accumulators
#pragma unroll
      for (int t6 = 0; t6 < 6; ++t6) {
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#define FX_OP(P_) __builtin_bit_cast(bf16x8v, (u32x4v){(P_)[0], (P_)[1], (P_)[2], (P_)[3]})
        const bf16x8v af = FX_OP(ap[b][PA[t6]]);
        acc[0][t6 & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, FX_OP(wreg[b][PB[t6]][0]), acc[0][t6 & 1], 0, 0, 0);
        acc[1][t6 & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, FX_OP(wreg[b][PB[t6]][1]), acc[1][t6 & 1], 0, 0, 0);
#undef FX_OP
      }
      // (interleaving the next block's split with these matrix instructions -- sched_group_barrier,
      // one MFMA : three VALU -- was measured SLOWER, 0.73 -> 0.83 us for this phase: an issue slot
      // between two back-to-back 16x16x32 instructions costs more than the split hides)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = b; i < CHX; i += NB)
        xn[i] = __builtin_amdgcn_raw_buffer_load_b128(xres, xoffv[i], xso, 0);
      // (waves 0 and 1 skip the early groups' loads: even an out-of-range load costs issue time)
      if (early) {
#pragma unroll
        for (int i = b; i < CHE; i += NB)
          xne[i] = __builtin_amdgcn_raw_buffer_load_b128(xres, xoffev[i], xso, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#undef FX_SPLIT
    // cross-wave reduction.  D layout 16x16: col = lane&15, row = 4*(lane>>4)+r
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        red[(wave * 16 + 4 * fq + r) * 33 + nt * 16 + fr] = acc[nt][0][r] + acc[nt][1][r];
    __syncthreads();
    TRACE_AT(0, 4);
    // the prefetched x becomes the current x HERE, before anything is stored: the wait for
    // it must not sit behind the write-through publish stores (see the header comment)
#pragma unroll
    for (int g = 0; g < CHX; ++g) {
      xc[g] = xn[g];
      asm volatile("" : "+v"(xc[g]));
    }
#pragma unroll
    for (int g = 0; g < CHE; ++g) {
      xce[g] = xne[g];
      asm volatile("" : "+v"(xce[g]));
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      accn[nt][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      accn[nt][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // waves 2 and 3: early groups of the NEXT step, beside the gate math of waves 0 and 1
    if (early && s + 1 < T) {
      if constexpr (XBF) {
#pragma unroll
        for (int b = 0; b < XBE; ++b) FX_GXE(b);
      } else {
#pragma unroll
        for (int g = 0; g < CHE; ++g) FX_GROUPF(xce[g], wxef[g], accn);
      }
    }

    if (owner) {
      float pre[4];
#pragma unroll
      for (int gte = 0; gte < 4; ++gte) {
        float v = bq[gte];
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[(w * 16 + bl) * 33 + gte * 8 + ul];
        pre[gte] = v;
      }
      const float g = pre[0];                 // linear candidate (ops.py:143)
      const float ig = sigmoid_hw(pre[1]);
      const float fg = sigmoid_hw(pre[2]);
      const float og = sigmoid_hw(pre[3]);
      c_state = ig * g + fg * c_state;        // ops.py:146
      const float h = og * tanh_hw(c_state);   // ops.py:147
      // (owner pointers walk the time axis: no 64-bit multiplies in front of the publish store)
      __hip_atomic_store(hp_w, h, RLX_AGENT);
      gs_w[0] = g; gs_w[H] = ig; gs_w[2 * H] = fg; gs_w[3 * H] = og;
      cs_w[0] = c_state;
      hp_w += hstep_w; gs_w += gstep_w; cs_w += cstep_w;
    }
    TRACE_AT(0, 5);
    __syncthreads();   // `red` reuse
    TRACE_AT(0, 7);
  }
#undef FX_GX
#undef FX_GXE
#undef FX_BLOCKX
#undef FX_GROUPF
#undef FX_OPX
}

// ---------------------------------------------------------------------------
// backward (BPTT), reduce-scatter form
// ---------------------------------------------------------------------------
// An all-gather formulation (every workgroup reads its cluster's whole da_t, 16 x 4H
// floats, each step: rounds 1-2) costs 441 us per cfg-2 launch against 303 us for this one.
// Here the product dh_{prev} = da_t Wh^T is split along K instead.  A cluster = 16 batch rows of one direction.  Producer group p
// owns U units, i.e. the 4U columns {gate*H + u0 + j} of da_t, keeps the matching
// Wh columns stationary IN REGISTERS and publishes its partial dh for all units
//     part_p[r][i] = sum_{n in own 4U columns} da_t[r][n] * Wh[i][n]
// The group is S "twin" workgroups that all redo the (cheap) exchange read + gate
// math of the group's 16 x U elements and each compute 1/S of the 16-unit output
// tiles, so the per-step MFMA chain is short and every MFMA row is a real batch
// row.  A group then reads only the P partials of its own U units, sums them in a
// fixed order (deterministic) and does the gate math.  da_t itself is no longer an
// exchange medium: twin 0 writes it with plain stores, no prefill needed.
//
// Hand-off: the payload is the flag, with no sentinel to restore: bit 0 of every
// published fp32 word carries the PHASE of its ring slot (use count & 1), i.e. a
// partial keeps a 23-bit mantissa (<= 1 ulp of one partial, below the fp32
// rounding of the P-term sum it feeds; NaNs stay NaNs).  A 16-byte chunk is valid
// when all four words carry the expected phase: stale words of the slot's previous
// use have the opposite phase, the host prefills phase 1 and the first use publishes
// phase 0.  Ring depth 3: a workgroup Y publishes step n (slot n%3) only after the
// twins owning ITS tiles published n-1 in every group, and each of those only after
// the twins owning THEIR tiles published n-2 -- every twin class owns some group's
// tile, so every workgroup has published n-2, i.e. finished reading step n-3 (all of
// its waves passed the block barrier behind those reads), before slot n%3 is rewritten.
// (Two slots are NOT enough with twins: a twin nobody in Y's tile class waits for
// may lag two steps.)
//
// MFMA orientation: D[unit][batch row] = W-tile(16 units x 4 k) * da^T(4 k x 16
// rows): each lane ends up with 4 CONSECUTIVE units of one batch row, and the ring
// is laid out [slot][cluster][producer][tile][row][16 units] so that one store
// instruction writes one contiguous 1 KB tile and a consumer reads whole rows.
struct LstmBwdRsArgs {
  const float* dy;
  const float* Wh[2];
  const float* gates[2];
  const float* cell[2];
  float* da[2];
  float* ring;
  int* status;
  int T, B, H, ndir, lddy, ldw, P, G, S, NT, NI, D, xmap;
  unsigned spin_limit;
  int fault;
  float* dbslab;   // optional [ndir*G][4H]: per-cluster bias-gradient partials (colsum of da)
};

// cache-policy bits of the reduce-scatter's publish stores: 16 = sc1 (agent scope, write-through);
// build variants -DRS_STORE_AUX=17 (sc0 sc1: system scope) / 18 (sc1 nt) exist for the PMC comparison
// in profiles/EXPERIMENTS.md -- WRITE_SIZE is the same for all three, the step is not
#ifndef RS_STORE_AUX
#define RS_STORE_AUX 16
#endif
__device__ __forceinline__ void store_sc1_b128(__amdgpu_buffer_rsrc_t r, unsigned off, v4u v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, RS_STORE_AUX);
}

#define RS_NI_MAX 6

template <int U, int NTW>
__global__ __launch_bounds__(512) void lstm_bwd_rs_kernel(LstmBwdRsArgs a) {
  constexpr int NW = 8;
  constexpr int KG = U / 4;          // 16-wide k-groups of the 4U own columns
  constexpr int OWN = 16 * U;        // (row, unit) elements of the group
  constexpr int CPP = 4 * U;         // 16-byte chunks per producer
  constexpr int PPR = 512 / CPP;     // producers covered by one round of loads
  constexpr int LDA = 4 * U + 4;
  __shared__ __attribute__((aligned(16))) float psum[PPR * OWN];  // 8 KB
  __shared__ __attribute__((aligned(16))) float atile[16 * LDA];  // da_t own columns [row][k]
  // OWNSPLIT (U = 32, the wide layers): da_t as the three bf16 piece planes in the matrix instruction's
  // operand order -- [piece][row][k-block][fq][8 k of the lane], rows padded by 16 bytes.  The OWNER of a
  // (row, unit) splits its four gate values once (two split_pair) and stores twelve half-words; every
  // wave then reads its B operands with three 16-byte LDS loads per k-block instead of splitting the fp32
  // tile itself (112 vector instructions per wave and step behind the second barrier at U = 32).
  // Measured: H = 600 BPTT 592 -> 544 us per launch beside the groups, cfg 4 as written 8.65 -> 8.49 ms;
  // at U = 16 (56 instructions per wave, twelve half-word stores against four word stores in the owner
  // phase) it is a wash to a loss -- 249 against 246 us alone, 320 against 317.5 in the step -- and
  // stays on the consumer-side split (profiles/r05_l_bptt_owner_split.txt).
  constexpr bool OWNSPLIT = (U >= 32);
  constexpr int PKB = (U / 4) / 2 > 0 ? (U / 4) / 2 : 1;
  constexpr int PRST = PKB * 32 + 8, PPST = 16 * PRST;           // half-words per row / per piece plane
  __shared__ __attribute__((aligned(16))) uint16_t ptile[OWNSPLIT ? 3 * PPST : 8];

  const int H = a.H, B = a.B, T = a.T, P = a.P, S = a.S;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bid = blockIdx.x;
  if (a.fault && bid == 0) return;   // test hook
  const int ncl = a.ndir * a.G;
  // xmap: consecutive block ids cycle over the clusters (so a cluster's workgroups
  // share their id modulo 8 = their XCD whenever ncl divides 8 or vice versa)
  int cl, p, tw;
  if (a.xmap == 2) {
    // twins on ONE XCD (workgroup id % 8 = XCD): the S twins of a group re-read the same
    // gates / cell / dy lines every step -- in one XCD's L2 the second and third read hit.
    // XCD x = cl + ncl * (p % npar), npar = 8 / ncl; slot j = (p / npar) * S + tw; bid = x + 8 j.
    // (grid = 8 * ceil(P / npar) * S: the few slots with p >= P exit at once)
    const int npar = 8 / ncl;
    const int x = bid & 7, j = bid >> 3;
    cl = x % ncl;
    tw = j % S;
    p = (j / S) * npar + x / ncl;
    if (p >= P) return;
  } else {
    cl = a.xmap ? bid % ncl : bid / (P * S);
    const int m = a.xmap ? bid / ncl : bid % (P * S);
    p = m / S; tw = m % S;
  }
  const int dir = cl / a.G, grp = cl % a.G;
  const int u0 = p * U, b0 = grp * 16;
  const int fr = lane & 15, fq = lane >> 4;

  // stationary weights: lane (fr, fq) of tile `tl` holds Wh[tl*16 + fr][own column
  // k0 .. k0+3], k0 = kg*16 + fq*4 (the same K permutation as the B operand below)
  // BF (U >= 16): the product runs on the bf16 matrix cores like the forward kernels' (six bf16 piece
  // products per fp32 product, split_pair): k-block b = k-groups 2b | 2b+1, the weight pieces
  // stationary, da_t split by every wave from the LDS tile.  Measured: U = 32 (H = 600) 661 -> 591 us per
  // launch, U = 16 (cfg 2) 352 -> 342 us in the step (300 alone); U = 8 keeps v_mfma_f32_16x16x4_f32.
  constexpr bool BF = (U >= 16);
  constexpr int KB = KG / 2;
  f32x4 wreg[BF ? 1 : NTW][BF ? 1 : KG];
  uint32_t wregp[BF ? NTW : 1][BF ? KB : 1][3][4];
  {
    const float* W = a.Wh[dir];
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
      const int tl = tw + S * (wave + NW * i);
      const int unit_i = tl * 16 + fr;
      f32x4 wf[KG];
#pragma unroll
      for (int kg = 0; kg < KG; ++kg) {
        const int k0 = kg * 16 + fq * 4;
        const int gate = k0 / U, j0 = k0 % U;
        f32x4 w = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (tl < a.NT && unit_i < H && u0 + j0 < H)
          w = *reinterpret_cast<const f32x4*>(&W[(size_t)unit_i * a.ldw + gate * H + u0 + j0]);
        wf[kg] = w;
      }
      if constexpr (BF) {
#pragma unroll
        for (int b = 0; b < KB; ++b) {
          split_pair(wf[2 * b][0], wf[2 * b][1], wregp[i][b][0][0], wregp[i][b][1][0], wregp[i][b][2][0]);
          split_pair(wf[2 * b][2], wf[2 * b][3], wregp[i][b][0][1], wregp[i][b][1][1], wregp[i][b][2][1]);
          split_pair(wf[2 * b + 1][0], wf[2 * b + 1][1], wregp[i][b][0][2], wregp[i][b][1][2], wregp[i][b][2][2]);
          split_pair(wf[2 * b + 1][2], wf[2 * b + 1][3], wregp[i][b][0][3], wregp[i][b][1][3], wregp[i][b][2][3]);
        }
      } else {
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) wreg[i][kg] = wf[kg];
      }
    }
  }

  const size_t slot_floats = (size_t)ncl * P * a.NT * 256;
  const unsigned rbytes = (unsigned)(slot_floats * a.D * sizeof(float));
  const __amdgpu_buffer_rsrc_t rres = make_rsrc(a.ring, rbytes);

  // exchange-read mapping: chunk = 4 units of one (producer, row); this thread
  // serves producers qq, qq+PPR, ... for a fixed (row, 4-unit quarter)
  const int qq = tid / CPP, within = tid % CPP;
  const int xr = within / (U / 4), xunit = u0 + (within % (U / 4)) * 4;
  const unsigned xoff = (unsigned)((((xunit >> 4) * 16 + xr) * 16 + (xunit & 15)) * 4);

  // gate-math ownership: threads 0..OWN-1 -> (row, unit)
  const bool othr = tid < OWN;
  const int orow = tid / U, oj = tid % U;
  const int bg = b0 + orow, unit = u0 + oj;
  const bool owner = othr && bg < B && unit < H;
  float dc_state = 0.f;
  float dbacc[4] = {0.f, 0.f, 0.f, 0.f};   // bias gradient: this (row, unit)'s da summed over time
  // where the owner's gate g lands in a piece plane: column c = g U + oj is k-group c / 16, position
  // c % 16 = 4 fq + j4 of it; the lane (row, fq) holds k-block b = (c / 16) / 2 as [half][j4]
  int powr[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int c = g * U + oj, kg = c >> 4, pos = c & 15;
    powr[g] = orow * PRST + ((kg >> 1) * 4 + (pos >> 2)) * 8 + (kg & 1) * 4 + (pos & 3);
  }

  // ring addressing without per-step multiplies or divisions: byte offsets inside a slot are fixed,
  // the slot and its phase are counters
  const unsigned slot_bytes = (unsigned)(slot_floats * sizeof(float));
  unsigned rd_off[RS_NI_MAX], rd_offv[RS_NI_MAX];
  bool rd_ok[RS_NI_MAX];
  // a load for a producer that does not exist (q >= P) is issued out of range and returns zeros on this
  // hardware (raw buffer: the range check covers the vector offset, the slot in the scalar offset is
  // outside it) -- but nothing DEPENDS on that: its words are masked out of the phase test and of the sum
  // (rd_any / rd_keep: an AND with a loop-invariant register where an OR / an AND with a constant stood,
  // the same instruction count; round 6, advisor finding)
  unsigned rd_any[RS_NI_MAX], rd_keep[RS_NI_MAX];
#pragma unroll
  for (int i = 0; i < RS_NI_MAX; ++i) {
    const int q = qq + PPR * i;
    rd_ok[i] = q < P;
    rd_off[i] = (unsigned)(((size_t)cl * P + q) * a.NT * 1024) + xoff;
    rd_offv[i] = rd_ok[i] ? rd_off[i] : rbytes;
    rd_any[i] = rd_ok[i] ? 0xFFFFFFFFu : 0u;
    rd_keep[i] = rd_ok[i] ? 0xFFFFFFFEu : 0u;
  }
  unsigned pub_off[NTW];
  bool pub_ok[NTW];
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
    const int tl = tw + S * (wave + NW * i);
    pub_ok[i] = tl < a.NT;
    pub_off[i] = (unsigned)((((size_t)cl * P + p) * a.NT + tl) * 1024) + (unsigned)((fr * 16 + fq * 4) * 4);
  }
  unsigned pslot = 0u, ppar = 0u, rslot = 0u, rpar = 0u;

  // the owner's saved activations of the current timestep: pointers that walk the time axis
  // (no 64-bit multiplies per step)
  const int tfirst = dir ? 0 : (T - 1);
  const ptrdiff_t tdir = dir ? 1 : -1;
  const ptrdiff_t gstep = tdir * (ptrdiff_t)B * (4 * H), cstep = tdir * (ptrdiff_t)B * H,
                  ystep = tdir * (ptrdiff_t)B * a.lddy, cprev_d = cstep;
  const size_t orow0 = owner ? ((size_t)tfirst * B + bg) : 0;
  const size_t ounit = owner ? unit : 0;
  const float* gp = a.gates[dir] + orow0 * (4 * H) + ounit;
  const float* cp = a.cell[dir] + orow0 * H + ounit;
  const float* dyp = a.dy + orow0 * a.lddy + (owner ? dir * H + unit : 0);
  float* dp = a.da[dir] + orow0 * (4 * H) + ounit;

  // wait for the NI partial sums of this thread's chunk published in ring slot `sbase`, phase `par`;
  // their sum goes to psum
  auto exchange = [&](auto ni_c, const unsigned sbase, const unsigned par, const int s) {
    (void)s;   // the trace build stamps by step
    constexpr int NI = decltype(ni_c)::value;
    v4u av[NI];
    unsigned spins = 0;
    for (;;) {
      // (rd_offv: the offset inside a slot, or out of range -> zeros; the slot in the scalar offset)
#pragma unroll
      for (int i = 0; i < NI; ++i) av[i] = load_sc1_b128s(rres, rd_offv[i], sbase);
      // every word must carry the expected phase in bit 0: one OR tree (phase 0) and one AND tree
      // (phase 1) over all words of the producers that exist, ONE test -- instead of a test and a
      // scalar AND per load
      unsigned orall = 0u, andall = 0xFFFFFFFFu;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        orall |= ((av[i][0] | av[i][1]) | (av[i][2] | av[i][3])) & rd_any[i];
        andall &= ((av[i][0] & av[i][1]) & (av[i][2] & av[i][3])) | ~rd_any[i];
      }
      const bool ok = par ? ((andall & 1u) != 0u) : ((orall & 1u) == 0u);
      if (__all(ok)) break;
      if (spin_fail(spins, a.status, lane, a.spin_limit)) break;
    }
    TRACE(1); TRACE_VAL(6, spins);
    f32x4 sum = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      v4u w = av[i];
      w[0] &= rd_keep[i]; w[1] &= rd_keep[i]; w[2] &= rd_keep[i]; w[3] &= rd_keep[i];   // phase bit off; no producer: 0
      sum += __builtin_bit_cast(f32x4, w);
    }
    *reinterpret_cast<f32x4*>(&psum[qq * OWN + within * 4]) = sum;
  };

  for (int s = 0; s < T; ++s) {
    const int t = dir ? s : (T - 1 - s);
    const int t_cprev = dir ? (t + 1) : (t - 1);

    TRACE(0);
    float gv[4] = {0.f, 0.f, 0.f, 0.f}, cv = 0.f, cpv = 0.f, dyv = 0.f;
    if (owner) {
      gv[0] = gp[0]; gv[1] = gp[H]; gv[2] = gp[2 * H]; gv[3] = gp[3 * H];
      cv = cp[0];
      if (t_cprev >= 0 && t_cprev < T) cpv = cp[cprev_d];
      dyv = dyp[0];
    }

    if (s > 0) {
      // NI (loads per thread and poll) is a launch constant: one switch per step onto code with
      // the count compiled in -- a runtime bound on every load, test and add was a chain of a
      // dozen scalar branches per poll
      switch (a.NI) {
        case 1: exchange(std::integral_constant<int, 1>{}, rslot * slot_bytes, rpar, s); break;
        case 2: exchange(std::integral_constant<int, 2>{}, rslot * slot_bytes, rpar, s); break;
        case 3: exchange(std::integral_constant<int, 3>{}, rslot * slot_bytes, rpar, s); break;
        case 4: exchange(std::integral_constant<int, 4>{}, rslot * slot_bytes, rpar, s); break;
        case 5: exchange(std::integral_constant<int, 5>{}, rslot * slot_bytes, rpar, s); break;
        default: exchange(std::integral_constant<int, 6>{}, rslot * slot_bytes, rpar, s); break;
      }
    }
    __syncthreads();
    TRACE(2);

    float dav[4] = {0.f, 0.f, 0.f, 0.f};
    if (othr) {
      float dh = dyv;
      if (s > 0) {
#pragma unroll
        for (int k = 0; k < PPR; ++k) dh += psum[k * OWN + tid];
      }
      if (owner) {
        const float g = gv[0], ig = gv[1], fg = gv[2], og = gv[3];
        const float tc = tanh_hw(cv);
        const float dc = dc_state + dh * og * (1.f - tc * tc);
        dav[0] = dc * ig;
        dav[1] = dc * g * ig * (1.f - ig);
        dav[2] = dc * cpv * fg * (1.f - fg);
        dav[3] = dh * tc * og * (1.f - og);
        dc_state = dc * fg;
        dbacc[0] += dav[0]; dbacc[1] += dav[1]; dbacc[2] += dav[2]; dbacc[3] += dav[3];
      }
      if constexpr (OWNSPLIT) {
        uint32_t ph[2], pm[2], pl[2];
        split_pair(dav[0], dav[1], ph[0], pm[0], pl[0]);
        split_pair(dav[2], dav[3], ph[1], pm[1], pl[1]);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const uint32_t hv = ph[g >> 1], mv = pm[g >> 1], lv = pl[g >> 1];
          uint16_t* pp = &ptile[powr[g]];
          pp[0] = (uint16_t)((g & 1) ? (hv >> 16) : (hv & 0xFFFFu));
          pp[PPST] = (uint16_t)((g & 1) ? (mv >> 16) : (mv & 0xFFFFu));
          pp[2 * PPST] = (uint16_t)((g & 1) ? (lv >> 16) : (lv & 0xFFFFu));
        }
      } else {
        float* ap = &atile[orow * LDA + oj];
        ap[0] = dav[0]; ap[U] = dav[1]; ap[2 * U] = dav[2]; ap[3 * U] = dav[3];
      }
    }
    __syncthreads();
    TRACE(3);

    if (s + 1 < T) {
      f32x4 bq[KG];
      if constexpr (!OWNSPLIT) {
#pragma unroll
        for (int kg = 0; kg < KG; ++kg)
          bq[kg] = *reinterpret_cast<const f32x4*>(&atile[fr * LDA + kg * 16 + fq * 4]);
      }
      // NACC independent accumulators per tile: a single dependent 16x16x4 chain
      // cannot issue back to back
      constexpr int NACC = 2;
      f32x4 acc2[NTW][NACC];
#pragma unroll
      for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int c = 0; c < NACC; ++c) acc2[i][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if constexpr (BF) {
#define BW_OP(P_) __builtin_bit_cast(bf16x8v, (u32x4v){(P_)[0], (P_)[1], (P_)[2], (P_)[3]})
#pragma unroll
        for (int b = 0; b < KB; ++b) {
          u32x4v bp[3];
          if constexpr (OWNSPLIT) {
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
              bp[pc] = *reinterpret_cast<const u32x4v*>(&ptile[pc * PPST + fr * PRST + (b * 4 + fq) * 8]);
          } else {
            uint32_t q[3][4];
            split_pair(bq[2 * b][0], bq[2 * b][1], q[0][0], q[1][0], q[2][0]);
            split_pair(bq[2 * b][2], bq[2 * b][3], q[0][1], q[1][1], q[2][1]);
            split_pair(bq[2 * b + 1][0], bq[2 * b + 1][1], q[0][2], q[1][2], q[2][2]);
            split_pair(bq[2 * b + 1][2], bq[2 * b + 1][3], q[0][3], q[1][3], q[2][3]);
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) bp[pc] = (u32x4v){q[pc][0], q[pc][1], q[pc][2], q[pc][3]};
          }
          // small terms first: (weight piece, da piece)
#pragma unroll
          for (int t6 = 0; t6 < 6; ++t6) {
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
            const bf16x8v bf = __builtin_bit_cast(bf16x8v, bp[PB[t6]]);
#pragma unroll
            for (int i = 0; i < NTW; ++i)
              acc2[i][t6 % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BW_OP(wregp[i][b][PA[t6]]), bf,
                                                                            acc2[i][t6 % NACC], 0, 0, 0);
          }
        }
#undef BW_OP
      } else {
#pragma unroll
        for (int kg = 0; kg < KG; ++kg)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < NTW; ++i)
              acc2[i][j % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                  wreg[i][kg][j], bq[kg][j], acc2[i][j % NACC], 0, 0, 0);
      }
      f32x4 acc[NTW];
#pragma unroll
      for (int i = 0; i < NTW; ++i) {
        acc[i] = acc2[i][0] + acc2[i][1];
        if (NACC == 4) acc[i] += acc2[i][2] + acc2[i][3];
      }
      TRACE(4);
      const unsigned ppub = ppar;
#pragma unroll
      for (int i = 0; i < NTW; ++i) {
        // (the slot stays in the VECTOR offset here: with it in the scalar offset the stores of
        // tiles that do not exist -- vector offset = buffer size -- were NOT dropped, B = 48: wrong dX)
        unsigned o = rbytes;   // out of range: dropped
        if (pub_ok[i]) o = pslot * slot_bytes + pub_off[i];
        v4u w = __builtin_bit_cast(v4u, acc[i]);
        w[0] = (w[0] & ~1u) | ppub; w[1] = (w[1] & ~1u) | ppub;
        w[2] = (w[2] & ~1u) | ppub; w[3] = (w[3] & ~1u) | ppub;
        store_sc1_b128(rres, o, w);
      }
    }
    // step s + 1 reads what step s published; the publishing slot walks the ring, its phase flips per lap
    rslot = pslot; rpar = ppar;
    if (++pslot == (unsigned)a.D) { pslot = 0u; ppar ^= 1u; }
    // da_t is only read by later kernels: plain stores, off the critical path
    if (owner && tw == 0) {
      dp[0] = dav[0]; dp[H] = dav[1]; dp[2 * H] = dav[2]; dp[3 * H] = dav[3];
    }
    gp += gstep; dp += gstep; cp += cstep; dyp += ystep;
    TRACE(5);
  }
  // partial db of the cluster (twin 0): the own 4U da columns summed over the 16 rows, so the
  // caller needs no column-sum kernels (6 launches per step at cfg 2, and a cross-stream join
  // at the tail of backward)
  if (a.dbslab != nullptr && tw == 0) {
    __syncthreads();
    if (othr) {
      float* ap = &atile[orow * LDA + oj];
      ap[0] = dbacc[0]; ap[U] = dbacc[1]; ap[2 * U] = dbacc[2]; ap[3 * U] = dbacc[3];
    }
    __syncthreads();
    if (tid < 4 * U) {
      float v = 0.f;
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) v += atile[rr * LDA + tid];
      const int gate = tid / U, j = tid % U;
      if (u0 + j < H) a.dbslab[(size_t)cl * (4 * H) + gate * H + u0 + j] = v;
    }
  }
}

// db_d[c] (+)= sum over the G clusters of direction d of their partial (fixed order: deterministic)
__global__ __launch_bounds__(256) void lstm_db_reduce_kernel(
    const float* __restrict__ dbslab, float* db0, float* db1, int G, int H4, float beta) {
  const int dir = blockIdx.y;
  float* db = dir ? db1 : db0;
  const int q4 = H4 / 4;
  for (int j = blockIdx.x * 256 + threadIdx.x; j < q4; j += gridDim.x * 256) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int g = 0; g < G; ++g)
      v += reinterpret_cast<const f32x4*>(dbslab + (size_t)(dir * G + g) * H4)[j];
    f32x4* o = reinterpret_cast<f32x4*>(db) + j;
    if (beta != 0.f) v += *o;
    *o = v;
  }
}

// ---------------------------------------------------------------------------
// prefill: every "not yet published" pattern / zero block of one call in ONE launch
// (hipMemsetAsync with a byte value lowers to a fill kernel PLUS copy kernels on this
// runtime, and each memset is its own node on the critical path of the layer)
// ---------------------------------------------------------------------------
// (FILL_MAXSEG, FillArgs, fill_share: csrc/common.h -- danet_center's one-launch kernel can carry a
// fill list as a rider, danet_encoder_prologue)
__global__ __launch_bounds__(256) void multi_fill_kernel(FillArgs a) {
  fill_share(a, (unsigned long long)blockIdx.x * 256 + threadIdx.x, (unsigned long long)gridDim.x * 256);
}

struct FillList {
  FillArgs a;
  FillList() { a.nseg = 0; }
  bool full() const { return a.nseg >= FILL_MAXSEG; }
  void add(void* p, size_t bytes, unsigned value) {
    a.ptr[a.nseg] = p; a.n16[a.nseg] = bytes / 16; a.value[a.nseg] = value; ++a.nseg;
  }
  hipError_t launch(hipStream_t stream) {
    unsigned long long tot = 0;
    for (int i = 0; i < a.nseg; ++i) tot += a.n16[i];
    unsigned long long blocks = (tot + 256 * 4 - 1) / (256 * 4);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    multi_fill_kernel<<<(unsigned)blocks, 256, 0, stream>>>(a);
    return hipGetLastError();
  }
};

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct LstmPlan { int MT, G, P, KP, NW, UN; size_t lds; };

// compute units of the current device (one persistent workgroup per CU must be co-resident);
// gfx950 = 256, also the fallback when no device is visible (size queries on a CPU-only host)
int dn_num_cus() {
  static std::atomic<int> n{0};          // cached once a device answered (idempotent: benign race)
  int c = n.load(std::memory_order_relaxed);
  if (c) return c;
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) == hipSuccess &&
      hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) {
    n.store(v, std::memory_order_relaxed);
    return v;
  }
  (void)hipGetLastError();
  return 256;
}

// MT=1 (16-row clusters) halves the per-step MFMA time and the payload per
// workgroup at the price of twice the workgroups; it is used whenever one
// workgroup per CU still fits.
static LstmPlan make_plan(int B, int H, int ndir) {
  LstmPlan pl;
  pl.UN = LSTM_UNITS_FWD;
  pl.P = cdiv(H, pl.UN);
  pl.KP = cdiv(H, 16) * 16;
  pl.MT = (ndir * cdiv(B, 16) * pl.P > dn_num_cus()) ? 2 : 1;
  // wide layers: 12 units per workgroup keep 16-row clusters on the GPU where 8 units
  // would need 32-row clusters (cfg 4 as written, H = 600: 200 workgroups of 16 rows instead of
  // 150 of 32: 5.8 -> 4.6 us per timestep, 12.15 -> 11.55 ms per cfg-4h600 step).
  // option lstm_fwd_un = 8|12 overrides.
  if (B > 4) {
    const int want = danet_opt(OPT_LSTM_FWD_UN);
    const bool fits12 = ndir * cdiv(B, 16) * cdiv(H, 12) <= dn_num_cus();
    if ((want == 12 || (want != 8 && pl.MT == 2)) && fits12) {
      pl.UN = 12; pl.P = cdiv(H, 12); pl.MT = 1;
    }
  }
  pl.G = cdiv(B, 16 * pl.MT);
  pl.NW = 4;             // one wave per SIMD
  // red [NW][16 MT][NCOL + 1] + the weight pieces of the k-blocks behind a wave's first three
  // ([NW][blocks - 3][3 pieces][NCOL / 16 column tiles][64 lanes] x 16 bytes; none up to H = 384)
  { const int ngw = cdiv(pl.KP / 16, pl.NW), nbw = (ngw + 1) / 2;
    pl.lds = (((size_t)pl.NW * 16 * pl.MT * (4 * pl.UN + 1) + 3) & ~(size_t)3) * sizeof(float) +
             (size_t)pl.NW * (nbw > 3 ? nbw - 3 : 0) * 3 * (4 * pl.UN / 16) * 64 * 16; }
  return pl;
}

// reduce-scatter BPTT geometry for U units per producer group; ok == false when
// it does not fit one workgroup per CU
struct RsPlan { bool ok; int U, G, P, S, NT, NTW, NI, D; size_t ring_bytes; };
static RsPlan make_rs_plan(int B, int H, int ndir, int U) {
  RsPlan r;
  r.U = U; r.G = cdiv(B, 16); r.P = cdiv(H, U);
  r.NT = cdiv(r.P * U, 16); r.D = 3;
  r.NI = cdiv(r.P, 512 / (4 * U));
  const int ncl = ndir * r.G;
  int smax = dn_num_cus() / (ncl * r.P);
  if (smax > r.NT) smax = r.NT;
  // twins: the fewest that leave at most two tiles per SIMD, else as many as fit
  r.S = smax;
  for (int sv = 1; sv <= smax; ++sv)
    if (cdiv(r.NT, sv) <= 8) { r.S = sv; break; }
  { const int es = danet_opt(OPT_LSTM_BWD_S);
    if (es >= 1 && es <= smax) r.S = es; }
  r.NTW = r.S > 0 ? cdiv(cdiv(r.NT, r.S), 8) : 99;
  r.ring_bytes = (size_t)r.D * ncl * r.P * r.NT * 1024;
  r.ok = (H % 4 == 0) && smax >= 1 && r.NTW <= 5 && r.NI <= RS_NI_MAX &&
         r.ring_bytes < 0xFFFFFFF0ull;
  return r;
}
// option lstm_bwd_u = 8|16|32 pins U
static RsPlan choose_rs_plan(int B, int H, int ndir) {
  RsPlan none; none.ok = false; none.ring_bytes = 0;
  const int pin = danet_opt(OPT_LSTM_BWD_U);
  // fewest MFMAs per SIMD and step (tiles of a workgroup spread over 4 SIMDs, 4U/4
  // k-steps each); ties go to the smaller U (shorter dependent chains).  Measured at
  // cfg 2: U=16,S=3 2.4 us/step; U=32,S=5 2.8; U=8,S=1 3.0 (all-gather kernel: 3.5)
  // Wide layers (H > 384): the host runs this launch beside a ~1 ms weight-gradient group (ops._overlap_dw:
  // always, with the groups on the bf16 matrix cores).  There the 152-workgroup geometry U = 16 without twins
  // leaves the group and dX more of the GPU than U = 32 with three twins (228 workgroups): cfg 4 as written
  // 8.25-8.28 -> 7.97-8.01 ms per step, three alternating pairs (groups 1 040 -> 830 us, dX 516 -> 460), although
  // the kernel itself is slower beside the group (575 against 548 us) and alone.
  if (!pin && H > 384) {
    const RsPlan r = make_rs_plan(B, H, ndir, 16);
    if (r.ok) return r;
  }
  RsPlan best = none;
  int best_cost = 1 << 30;
  for (int U = 8; U <= 32; U *= 2) {
    if (pin && pin != U) continue;
    RsPlan r = make_rs_plan(B, H, ndir, U);
    if (!r.ok) continue;
    const int cost = cdiv(cdiv(r.NT, r.S), 4) * U;
    // (wide layers, H = 600: beside a 1 ms weight-gradient group the 152-workgroup geometry
    // U = 16 / no twins suffers less than U = 32 / 3 twins (14.4 vs 16.5 ms per cfg-4h600 step),
    // but the caller now runs those groups serially there (13.7 ms), where the faster-alone
    // U = 32 geometry wins: 4.8 vs 7.6 us per step.  DANET_LSTM_BWD_U pins U.)
    if (cost < best_cost) { best = r; best_cost = cost; }
  }
  return best;
}
static size_t ring_offset(int T) { return (64 + TRACE_BYTES(T) + 255) / 256 * 256; }
// DANET_LSTM_SPIN_LIMIT overrides the wait bound, DANET_LSTM_FAULT_INJECT=1 makes workgroup 0
// of every launch exit without publishing (tests of the timeout path only)
static unsigned spin_limit_env() {
  const long v = danet_opt(OPT_LSTM_SPIN_LIMIT);
  return v >= 256 ? (unsigned)v : SPIN_LIMIT;
}
static int fault_env() { return danet_opt(OPT_LSTM_FAULT_INJECT) == 1; }

size_t dn_ws_lstm(int T, int B, int H, int ndir) {
  // status word (+ padding) (+ trace records) + partial-dh ring of the BPTT kernel
  size_t ring = 0;
  for (int U = 8; U <= 32; U *= 2) {
    RsPlan r = make_rs_plan(B, H, ndir, U);
    if (r.ok && r.ring_bytes > ring) ring = r.ring_bytes;
  }
  // + per-cluster bias-gradient partials of danet_lstm_bwd_db
  return align_up(ring_offset(T) + ring, 256) + align_up((size_t)ndir * cdiv(B, 16) * 4 * H * sizeof(float), 256);
}

static int lstm_check_common(int T, int B, int H, int ndir, void* ws, size_t ws_bytes) {
  DANET_CHECK_ARG(T > 0 && B > 0 && H > 0, "lstm: non-positive shape");
  DANET_CHECK_ARG(ndir == 1 || ndir == 2, "lstm: ndir must be 1 or 2");
  if (H % 4 != 0) {
    danet_set_error("lstm: H=%d must be a multiple of 4", H);
    return DANET_ERR_UNSUPPORTED;
  }
  if (!ws || ((uintptr_t)ws & 15) != 0 || ws_bytes < dn_ws_lstm(T, B, H, ndir)) {
    danet_set_error("lstm: workspace too small or not 16-B aligned");
    return DANET_ERR_WORKSPACE;
  }
  return DANET_OK;
}

// ---- prefill of several launches' buffers in ONE fill launch ----------------------------------
// Every forward launch needs its output buffer prefilled with the "not yet published" sentinel
// (+ zero pad blocks), every reduce-scatter BPTT launch its partial-dh ring prefilled with phase
// 1.  Inside the entry points that is one fill launch per call (~6 us each: 6 per cfg-2 train
// step); a host that allocates the buffers of all layers up front prefills them with ONE call
// here and passes DANET_LSTM_PREFILLED to the launches.
static int prefill_fwd_segments(FillList& fl, hipStream_t stream, int T, int B, int ldy, int n,
                                float* const* ypads, void* const* wss) {
  const size_t blk = (size_t)B * ldy * sizeof(float);
  for (int i = 0; i < n; ++i) {
    DANET_CHECK_ARG(ypads[i] && ((uintptr_t)ypads[i] & 15) == 0, "lstm_prefill: ypad %d null / misaligned", i);
    if (fl.a.nseg + 4 > FILL_MAXSEG) { DANET_CHECK_HIP(fl.launch(stream)); fl = FillList(); }
    if (wss && wss[i]) fl.add(wss[i], 64 + TRACE_BYTES(T), 0u);
    fl.add((char*)ypads[i] + blk, (size_t)T * blk, SENTINEL);
    fl.add(ypads[i], blk, 0u);
    fl.add((char*)ypads[i] + (size_t)(T + 1) * blk, blk, 0u);
  }
  return DANET_OK;
}

static int prefill_bwd_segments(FillList& fl, hipStream_t stream, int T, const RsPlan& rs, int n,
                                void* const* wss) {
  for (int i = 0; i < n; ++i) {
    DANET_CHECK_ARG(wss[i] && ((uintptr_t)wss[i] & 15) == 0, "lstm_prefill: ws %d null / misaligned", i);
    if (fl.a.nseg + 2 > FILL_MAXSEG) { DANET_CHECK_HIP(fl.launch(stream)); fl = FillList(); }
    fl.add(wss[i], 64 + TRACE_BYTES(T), 0u);
    fl.add((char*)wss[i] + ring_offset(T), rs.ring_bytes, 1u);   // phase 1 in bit 0 of every word
  }
  return DANET_OK;
}

extern "C" int danet_lstm_fwd_prefill(danet_stream_t stream_, int T, int B, int ldy, int n,
                                      float* const* ypads, void* const* wss) {
  hipStream_t stream = (hipStream_t)stream_;
  DANET_CHECK_ARG(T > 0 && B > 0 && ldy > 0 && ldy % 4 == 0 && n > 0 && ypads, "lstm_fwd_prefill: bad argument");
  FillList fl;
  const int rc = prefill_fwd_segments(fl, stream, T, B, ldy, n, ypads, wss);
  if (rc) return rc;
  if (fl.a.nseg) DANET_CHECK_HIP(fl.launch(stream));
  return DANET_OK;
}

// A train step knows at the head of its forward pass that a backward pass follows: the forward
// launches' buffers AND the BPTT launches' rings in one fill launch (the rings are 4 MB each at cfg 2;
// the 6-us launch in front of the first BPTT kernel leaves the critical path).
extern "C" int danet_lstm_train_prefill(danet_stream_t stream_, int T, int B, int H, int ndir, int ldy,
                                        int n, float* const* ypads, void* const* fwd_wss,
                                        void* const* bwd_wss) {
  hipStream_t stream = (hipStream_t)stream_;
  DANET_CHECK_ARG(T > 0 && B > 0 && H > 0 && (ndir == 1 || ndir == 2) && ldy > 0 && ldy % 4 == 0 && n > 0 &&
                  ypads && bwd_wss, "lstm_train_prefill: bad argument");
  const RsPlan rs = choose_rs_plan(B, H, ndir);
  if (!rs.ok) {
    danet_set_error("lstm_train_prefill: B=%d H=%d outside the reduce-scatter geometry", B, H);
    return DANET_ERR_UNSUPPORTED;
  }
  FillList fl;
  int rc = prefill_fwd_segments(fl, stream, T, B, ldy, n, ypads, fwd_wss);
  if (rc) return rc;
  rc = prefill_bwd_segments(fl, stream, T, rs, n, bwd_wss);
  if (rc) return rc;
  if (fl.a.nseg) DANET_CHECK_HIP(fl.launch(stream));
  return DANET_OK;
}

// The head of the encoder (app/modules.py:209-223) in ONE launch: the input's mean-centre
// (danet_center) with the prefill of every recurrent launch that follows as a rider -- the centring
// kernel's workgroups wait for each other's partial sums anyway, three of their four waves stream the
// fill meanwhile (csrc/pointwise.hip).  bwd_wss == NULL: no backward pass follows (inference).  Falls
// back to danet_center + one fill launch where the one-launch centring form is not taken.
extern "C" int danet_encoder_prologue(danet_stream_t stream_, int B, int T, int D, const float* in,
                                      int in_layout, int ld_in, float* out, int out_layout, int ld_out,
                                      float* mean, int H, int ndir, int ldy, int n, float* const* ypads,
                                      void* const* fwd_wss, void* const* bwd_wss) {
  hipStream_t stream = (hipStream_t)stream_;
  DANET_CHECK_ARG(T > 0 && B > 0 && H > 0 && (ndir == 1 || ndir == 2) && ldy > 0 && ldy % 4 == 0 && n > 0 &&
                  ypads, "encoder_prologue: bad argument");
  FillList fl;
  int rc;
  const bool one = 4 * n + (bwd_wss ? 2 * n : 0) <= FILL_MAXSEG;   // else: the list's own launches
  if (bwd_wss) {
    const RsPlan rs = choose_rs_plan(B, H, ndir);
    if (!rs.ok) {
      danet_set_error("encoder_prologue: B=%d H=%d outside the reduce-scatter geometry", B, H);
      return DANET_ERR_UNSUPPORTED;
    }
    rc = prefill_fwd_segments(fl, stream, T, B, ldy, n, ypads, fwd_wss);
    if (rc) return rc;
    rc = prefill_bwd_segments(fl, stream, T, rs, n, bwd_wss);
    if (rc) return rc;
  } else {
    rc = prefill_fwd_segments(fl, stream, T, B, ldy, n, ypads, fwd_wss);
    if (rc) return rc;
  }
  bool taken = false;
  rc = dn_center(stream, B, T, D, in, in_layout, ld_in, out, out_layout, ld_out, mean,
                 one && fl.a.nseg ? &fl.a : nullptr, &taken);
  if (rc) return rc;
  if (!taken && fl.a.nseg) DANET_CHECK_HIP(fl.launch(stream));
  return DANET_OK;
}

extern "C" int danet_lstm_fwd(danet_stream_t stream_, int T, int B, int H, int ndir,
                              const float* gx_f, const float* gx_b,
                              const float* Wh_f, const float* Wh_b, int ldw,
                              float* ypad, int ldy, float* gates_f, float* gates_b,
                              float* cell_f, float* cell_b, void* ws, size_t ws_bytes,
                              int32_t* status, int flags) {
  hipEvent_t ev_start, ev_stop;
  dn_take_launch_events(&ev_start, &ev_stop);
  hipStream_t stream = (hipStream_t)stream_;
  int rc = lstm_check_common(T, B, H, ndir, ws, ws_bytes);
  if (rc) return rc;
  DANET_CHECK_ARG(gx_f && Wh_f && ypad && gates_f && cell_f, "lstm_fwd: null pointer");
  DANET_CHECK_ARG(ndir == 1 || (gx_b && Wh_b && gates_b && cell_b), "lstm_fwd: null bwd pointer");
  DANET_CHECK_ARG(ldy >= ndir * H && ldy % 4 == 0 && ldw >= 4 * H, "lstm_fwd: bad ld");
  DANET_CHECK_ARG(((uintptr_t)ypad & 15) == 0, "lstm_fwd: ypad must be 16-B aligned");
  DANET_CHECK_ARG((size_t)(T + 2) * B * ldy * 4 < 0xFFFFFFF0ull, "lstm_fwd: ypad > 4 GiB");
  LstmPlan pl = make_plan(B, H, ndir);
  if (pl.lds > 160 * 1024) {
    danet_set_error("lstm_fwd: H=%d needs %zu B LDS", H, pl.lds);
    return DANET_ERR_UNSUPPORTED;
  }
  const int nblk = ndir * pl.G * pl.P;
  if (nblk > dn_num_cus()) {  // 1 workgroup per CU must be co-resident
    danet_set_error("lstm_fwd: %d workgroups exceed the %d CUs", nblk, dn_num_cus());
    return DANET_ERR_UNSUPPORTED;
  }
  LstmFwdArgs a;
  a.gx[0] = gx_f; a.gx[1] = gx_b; a.Wh[0] = Wh_f; a.Wh[1] = Wh_b;
  a.gates[0] = gates_f; a.gates[1] = gates_b; a.cell[0] = cell_f; a.cell[1] = cell_b;
  a.ypad = ypad; a.status = status ? (int*)status : (int*)ws;
  a.spin_limit = spin_limit_env(); a.fault = fault_env();
  a.T = T; a.B = B; a.H = H; a.ndir = ndir; a.ldy = ldy; a.ldw = ldw;
  a.P = pl.P; a.G = pl.G; a.KP = pl.KP;
  a.xmap = (danet_opt(OPT_LSTM_XMAP) >= 0 ? danet_opt(OPT_LSTM_XMAP) : 1);
  // status word; "not yet published" sentinel in the T interior blocks of ypad; the
  // zero initial state in pad blocks 0 and T+1 (main.py:108-123) -- one launch
  const size_t blk = (size_t)B * ldy * sizeof(float);
  if (!(flags & DANET_LSTM_PREFILLED)) {
    FillList fl;
    fl.add(ws, 64 + TRACE_BYTES(T), 0u);
    fl.add((char*)ypad + blk, (size_t)T * blk, SENTINEL);
    fl.add(ypad, blk, 0u);
    fl.add((char*)ypad + (size_t)(T + 1) * blk, blk, 0u);
    DANET_CHECK_HIP(fl.launch(stream));
  }
#define LAUNCH_FWD(MTV, NWV, UNV)                                                    \
  do {                                                                               \
    DANET_CHECK_HIP(hipFuncSetAttribute((const void*)lstm_fwd_kernel<MTV, NWV, UNV>,  \
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds));                   \
    DN_LAUNCH_EV((lstm_fwd_kernel<MTV, NWV, UNV>), nblk, 64 * NWV, pl.lds, stream, a); \
  } while (0)
  // tiny batch (the B = 1 demo / inference path): GEMV on the vector ALU instead of 1/16-used
  // MFMA tiles.  DANET_LSTM_FWD_SMALL=0 keeps the MFMA kernel.
  if (B <= 4 && H <= 16 * FWD_CH * 4 && danet_opt(OPT_LSTM_FWD_SMALL) != 0) {
    const size_t lds = ((size_t)pl.KP * 32 + (size_t)4 * 4 * 33) * sizeof(float);
    DANET_CHECK_HIP(hipFuncSetAttribute((const void*)lstm_fwd_small_kernel<4>,
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    DANET_CHECK_HIP(hipFuncSetAttribute((const void*)lstm_fwd_small_kernel<1>,
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (B == 1) DN_LAUNCH_EV(lstm_fwd_small_kernel<1>, ndir * pl.P, 256, lds, stream, a);
    else DN_LAUNCH_EV(lstm_fwd_small_kernel<4>, ndir * pl.P, 256, lds, stream, a);
    DANET_CHECK_LAUNCH();
    return DANET_OK;
  }
  if (pl.UN == 12) LAUNCH_FWD(1, 4, 12);
  else if (pl.MT == 2) LAUNCH_FWD(2, 4, 8);
  else LAUNCH_FWD(1, 4, 8);
  DANET_CHECK_LAUNCH();
  return DANET_OK;
}

// Fused forward: envelope check shared by the query and the launch
static bool fwd_fused_ok(int T, int B, int H, int ndir, int D, int* CHX) {
  if (T <= 0 || B <= 0 || H <= 0 || D <= 0 || (ndir != 1 && ndir != 2)) return false;
  if (H % 4 != 0 || H > 16 * FWD_CH * 4 || D > 640) return false;
  const int P = cdiv(H, LSTM_UNITS_FWD), G = cdiv(B, 16);
  if (ndir * G * P > dn_num_cus()) return false;
  // DANET_LSTM_FWD_FUSED=0 turns the path off, =1 forces it inside the envelope; otherwise it is
  // used where it beats the hoisted GEMM: the input-half MFMAs add ~1.0 us * D/600 to a step
  // (measured, profiles/r02_d_fused_fwd_trace.txt), the hoisted GEMM costs ~1.15 us * B/32 *
  // D/600 per step (80 TFLOP/s) -- i.e. from about B >= 24 on; B = 1 inference (cfg 5) stays
  // on the hoisted path (2.66 vs 1.84 us per step).
  { const int e = danet_opt(OPT_LSTM_FWD_FUSED);
    if (e == 0) return false;
    if (e != 1 && B < 24) return false; }
  // k-groups: 4 * window + 2 * early >= D / 16  (see lstm_fwd_fx_kernel)
  if (CHX) *CHX = D <= 160 ? 2 : (D <= 320 ? 4 : (D <= 608 ? 8 : 9));
  return true;
}

extern "C" int danet_lstm_fwd_fused_supported(int T, int B, int H, int ndir, int D) {
  return fwd_fused_ok(T, B, H, ndir, D, nullptr) ? 1 : 0;
}

extern "C" int danet_lstm_fwd_fused(danet_stream_t stream_, int T, int B, int H, int ndir,
                                    const float* x, int ldx, int D,
                                    const float* W_f, const float* W_b, int ldw,
                                    const float* bias_f, const float* bias_b,
                                    float* ypad, int ldy, float* gates_f, float* gates_b,
                                    float* cell_f, float* cell_b, void* ws, size_t ws_bytes,
                                    int32_t* status, int flags) {
  hipEvent_t ev_start, ev_stop;
  dn_take_launch_events(&ev_start, &ev_stop);
  hipStream_t stream = (hipStream_t)stream_;
  int rc = lstm_check_common(T, B, H, ndir, ws, ws_bytes);
  if (rc) return rc;
  DANET_CHECK_ARG(x && W_f && bias_f && ypad && gates_f && cell_f, "lstm_fwd_fused: null pointer");
  DANET_CHECK_ARG(ndir == 1 || (W_b && bias_b && gates_b && cell_b), "lstm_fwd_fused: null bwd pointer");
  DANET_CHECK_ARG(D > 0 && ldx >= D && ldx % 4 == 0 && ((uintptr_t)x & 15) == 0,
                  "lstm_fwd_fused: x must be 16-B aligned with ldx %% 4 == 0, ldx >= D");
  DANET_CHECK_ARG(ldy >= ndir * H && ldy % 4 == 0 && ldw >= 4 * H, "lstm_fwd_fused: bad ld");
  DANET_CHECK_ARG(((uintptr_t)ypad & 15) == 0, "lstm_fwd_fused: ypad must be 16-B aligned");
  DANET_CHECK_ARG((size_t)(T + 2) * B * ldy * 4 < 0xFFFFFFF0ull &&
                  (size_t)T * B * ldx * 4 < 0xFFFFFFF0ull, "lstm_fwd_fused: tensor > 4 GiB");
  int CHX = 0;
  if (!fwd_fused_ok(T, B, H, ndir, D, &CHX)) {
    danet_set_error("lstm_fwd_fused: T=%d B=%d H=%d D=%d outside the fused envelope "
                    "(H <= 320, D <= 640, one workgroup per CU)", T, B, H, D);
    return DANET_ERR_UNSUPPORTED;
  }
  LstmFwdFxArgs a;
  a.x = x; a.W[0] = W_f; a.W[1] = W_b; a.bias[0] = bias_f; a.bias[1] = bias_b;
  a.gates[0] = gates_f; a.gates[1] = gates_b; a.cell[0] = cell_f; a.cell[1] = cell_b;
  a.ypad = ypad; a.status = status ? (int*)status : (int*)ws; a.status_ws = ws;
  a.spin_limit = spin_limit_env(); a.fault = fault_env();
  a.T = T; a.B = B; a.H = H; a.D = D; a.ndir = ndir; a.ldx = ldx; a.ldy = ldy; a.ldw = ldw;
  a.P = cdiv(H, LSTM_UNITS_FWD); a.G = cdiv(B, 16); a.KP = cdiv(H, 16) * 16;
  a.DP = cdiv(D, 16) * 16;
  a.xmap = (danet_opt(OPT_LSTM_XMAP) >= 0 ? danet_opt(OPT_LSTM_XMAP) : 1);
  const size_t lds = ((size_t)a.KP * 32 + (size_t)4 * 16 * 33) * sizeof(float);
  const size_t blk = (size_t)B * ldy * sizeof(float);
  if (!(flags & DANET_LSTM_PREFILLED)) {
    FillList fl;
    fl.add(ws, 64 + TRACE_BYTES(T), 0u);
    fl.add((char*)ypad + blk, (size_t)T * blk, SENTINEL);
    fl.add(ypad, blk, 0u);
    fl.add((char*)ypad + (size_t)(T + 1) * blk, blk, 0u);
    DANET_CHECK_HIP(fl.launch(stream));
  }
  const int nblk = ndir * a.G * a.P;
#define FX_LAUNCH(W_, E_)                                                                \
  do {                                                                                   \
    DANET_CHECK_HIP(hipFuncSetAttribute((const void*)lstm_fwd_fx_kernel<W_, E_>,         \
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                          \
    DN_LAUNCH_EV((lstm_fwd_fx_kernel<W_, E_>), nblk, 256, lds, stream, a);               \
  } while (0)
  // the early groups must fit the gate phase (~0.5 us = 4-5 groups), the window groups the
  // exchange wait (~1 us = 9 groups); measured at D = 600: 7 + 6 loses 0.27 us per step at the
  // closing barrier, 8 + 3 has slack on both sides
  if (CHX == 2) FX_LAUNCH(2, 1);          // D <= 160: 8 + 2 k-groups
  else if (CHX == 4) FX_LAUNCH(4, 2);     // D <= 320: 16 + 4
  else if (CHX == 8) FX_LAUNCH(8, 3);     // D <= 608: 32 + 6
  else FX_LAUNCH(8, 4);                   // D <= 640: 32 + 8
#undef FX_LAUNCH
  DANET_CHECK_LAUNCH();
  return DANET_OK;
}

extern "C" int danet_lstm_bwd_db_supported(int T, int B, int H, int ndir) {
  if (T <= 0 || B <= 0 || H <= 0 || H % 4 != 0 || (ndir != 1 && ndir != 2)) return 0;
  return choose_rs_plan(B, H, ndir).ok ? 1 : 0;
}

// The per-cluster bias-gradient partials a danet_lstm_bwd(..., DANET_LSTM_DB_DEFERRED) launch left
// in its workspace, summed into db: 6 us the caller can put on any stream ordered behind that launch
// instead of between the BPTT kernel and the dX product that waits for it.
extern "C" int danet_lstm_bwd_db_reduce(danet_stream_t stream_, int T, int B, int H, int ndir,
                                        float* db_f, float* db_b, float beta,
                                        const void* ws, size_t ws_bytes) {
  DANET_CHECK_ARG(db_f && (ndir == 1 || db_b), "lstm_bwd_db_reduce: null db pointer");
  DANET_CHECK_ARG(beta == 0.f || beta == 1.f, "lstm_bwd_db_reduce: beta must be 0 or 1");
  if (!danet_lstm_bwd_db_supported(T, B, H, ndir)) {
    danet_set_error("lstm_bwd_db_reduce: B=%d H=%d outside the reduce-scatter geometry", B, H);
    return DANET_ERR_UNSUPPORTED;
  }
  int rc = lstm_check_common(T, B, H, ndir, const_cast<void*>(ws), ws_bytes);
  if (rc) return rc;
  const RsPlan rs = choose_rs_plan(B, H, ndir);
  const float* slab = (const float*)((const char*)ws + align_up(ring_offset(T) + rs.ring_bytes, 256));
  dim3 grid((unsigned)cdiv(H, 256), ndir);
  lstm_db_reduce_kernel<<<grid, 256, 0, (hipStream_t)stream_>>>(slab, db_f, db_b, rs.G, 4 * H, beta);
  DANET_CHECK_LAUNCH();
  return DANET_OK;
}

extern "C" int danet_lstm_bwd(danet_stream_t stream_, int T, int B, int H, int ndir,
                              const float* dy, int lddy,
                              const float* Wh_f, const float* Wh_b, int ldw,
                              const float* gates_f, const float* gates_b,
                              const float* cell_f, const float* cell_b,
                              float* da_f, float* da_b, float* db_f, float* db_b, float beta,
                              void* ws, size_t ws_bytes, int32_t* status, int flags) {
  hipEvent_t ev_start, ev_stop;
  dn_take_launch_events(&ev_start, &ev_stop);
  if (db_f) {
    DANET_CHECK_ARG(ndir == 1 || db_b, "lstm_bwd: null db pointer");
    DANET_CHECK_ARG((((uintptr_t)db_f | (uintptr_t)db_b) & 15) == 0, "lstm_bwd: db must be 16-B aligned");
    DANET_CHECK_ARG(beta == 0.f || beta == 1.f, "lstm_bwd: beta must be 0 or 1");
    if (!danet_lstm_bwd_db_supported(T, B, H, ndir)) {
      danet_set_error("lstm_bwd: db requested, but B=%d H=%d is outside the reduce-scatter geometry", B, H);
      return DANET_ERR_UNSUPPORTED;
    }
  } else {
    DANET_CHECK_ARG((flags & DANET_LSTM_DB_DEFERRED) == 0, "lstm_bwd: DB_DEFERRED without db");
  }
  hipStream_t stream = (hipStream_t)stream_;
  int rc = lstm_check_common(T, B, H, ndir, ws, ws_bytes);
  if (rc) return rc;
  DANET_CHECK_ARG(dy && Wh_f && gates_f && cell_f && da_f, "lstm_bwd: null pointer");
  DANET_CHECK_ARG(ndir == 1 || (Wh_b && gates_b && cell_b && da_b), "lstm_bwd: null bwd pointer");
  DANET_CHECK_ARG(lddy >= ndir * H && ldw >= 4 * H, "lstm_bwd: bad ld");
  DANET_CHECK_ARG(((uintptr_t)da_f & 15) == 0 && ((uintptr_t)da_b & 15) == 0,
                  "lstm_bwd: da must be 16-B aligned");
  DANET_CHECK_ARG((size_t)T * B * 4 * H * 4 < 0xFFFFFFF0ull, "lstm_bwd: da > 4 GiB");
  const RsPlan rs = choose_rs_plan(B, H, ndir);
  if (rs.ok) {
    LstmBwdRsArgs a;
    a.dy = dy; a.lddy = lddy; a.Wh[0] = Wh_f; a.Wh[1] = Wh_b; a.ldw = ldw;
    a.gates[0] = gates_f; a.gates[1] = gates_b; a.cell[0] = cell_f; a.cell[1] = cell_b;
    a.da[0] = da_f; a.da[1] = da_b; a.status = status ? (int*)status : (int*)ws;
    a.spin_limit = spin_limit_env(); a.fault = fault_env();
    a.ring = (float*)((char*)ws + ring_offset(T));
    a.T = T; a.B = B; a.H = H; a.ndir = ndir; a.P = rs.P; a.G = rs.G; a.S = rs.S;
    a.NT = rs.NT; a.NI = rs.NI; a.D = rs.D;
    a.xmap = (danet_opt(OPT_LSTM_XMAP) >= 0 ? danet_opt(OPT_LSTM_XMAP) : 1);
    a.dbslab = db_f ? (float*)((char*)ws + align_up(ring_offset(T) + rs.ring_bytes, 256)) : nullptr;
    if (!(flags & DANET_LSTM_PREFILLED)) {
      FillList fl;
      fl.add(ws, 64 + TRACE_BYTES(T), 0u);
      fl.add(a.ring, rs.ring_bytes, 1u);   // phase 1 in bit 0 of every word
      DANET_CHECK_HIP(fl.launch(stream));
    }
    int nblk = ndir * rs.G * rs.P * rs.S;
    {
      // twin-co-located order (xmap 2) where the clusters divide the 8 XCDs and the padded grid
      // still fits one workgroup per CU
      const int ncl = ndir * rs.G;
      if (a.xmap == 2) a.xmap = 1;
      if (danet_opt(OPT_LSTM_BWD_TWIN_XCD) == 1 && a.xmap == 1 && ncl <= 8 && 8 % ncl == 0 && rs.S > 1) {
        const int npar = 8 / ncl;
        const int g2 = 8 * cdiv(rs.P, npar) * rs.S;
        if (g2 <= dn_num_cus()) { a.xmap = 2; nblk = g2; }
      }
    }
#define LAUNCH_RS(UV, NTWV) DN_LAUNCH_EV((lstm_bwd_rs_kernel<UV, NTWV>), nblk, 512, 0, stream, a)
#define LAUNCH_RS_U(UV)                                          \
    switch (rs.NTW) {                                            \
      case 1: LAUNCH_RS(UV, 1); break; case 2: LAUNCH_RS(UV, 2); break; \
      case 3: LAUNCH_RS(UV, 3); break; case 4: LAUNCH_RS(UV, 4); break; \
      default: LAUNCH_RS(UV, 5); break; }
    if (rs.U == 8) { LAUNCH_RS_U(8) } else if (rs.U == 16) { LAUNCH_RS_U(16) } else { LAUNCH_RS_U(32) }
    DANET_CHECK_LAUNCH();
    if (db_f && !(flags & DANET_LSTM_DB_DEFERRED)) {
      dim3 grid((unsigned)cdiv(H, 256), ndir);
      lstm_db_reduce_kernel<<<grid, 256, 0, stream>>>(a.dbslab, db_f, db_b, rs.G, 4 * H, beta);
      DANET_CHECK_LAUNCH();
    }
    return DANET_OK;
  }
  danet_set_error("lstm_bwd: B=%d H=%d ndir=%d outside the reduce-scatter geometry (one workgroup per CU)",
                  B, H, ndir);
  return DANET_ERR_UNSUPPORTED;
}
