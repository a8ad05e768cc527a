// GMM patch prior on gfx950 (CDNA4).
//
// Forward: for every overlapping 8x8 patch x (mean subtracted) and every mixture component k
//     y_k = x^T P_k - m_k ,  q_k = sum_j w_j y_kj^2 ,  l_k = c_k - q_k / 2 ,  v = max_k l_k | logsumexp_k l_k
// (jolideco/priors/patches/gmm.py:262-281, priors/patches/core.py:189-246).  Per component this is a
// 64 x 64 matrix applied to every patch: a dense contraction, FLOP-bound on the fp32 roof.  It runs on
// the exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32: bit-for-bit an fmaf chain in pixel order):
//   M = whitened coordinate j (four 16-blocks), N = patch (16 per MFMA), K = pixel (4 per MFMA).
//   * P_k = (L_k^-1)^T is UPPER TRIANGULAR (jolideco/utils/numpy.py:16-34), so y_j only needs pixels
//     i <= j: 16-block jb of the whitened coordinates needs pixel steps 0 .. 4 (jb + 1) - 1.  Skipping
//     the all-zero blocks removes 24 of the 64 MFMAs per (component, 16 patches) and changes no bit of
//     the result (the skipped terms are exact zeros at the END of each fmaf chain).  jd_gmm_create checks
//     the structure; a non-triangular matrix set takes the dense variant of the same kernel;
//   * A operand = P'_k = P_k diag(sqrt w) fragments (pixel weights folded into the columns, fragment
//     order prepared once on the host), streamed from L2, register double-buffered across components;
//   * B operand = mean-subtracted patches, staged once per block in LDS in fragment order;
//   * the accumulators start at -m'_k so the mean shift costs nothing;
//   * C layout puts the patch on the lane (n = lane & 15) and the whitened coordinate in the registers,
//     so sum_j y_j^2 is an in-lane sum + two VALU lane swaps; (Np, K) never leaves the CU.
// One block = 4 waves (one per SIMD) shares TB tiles of 32 patches and splits the K components four
// ways; the partial (max, arg-max) | (max, sum-exp) results are merged through LDS in component order.
// Backward (max mode): the patches are bucketed by arg-max component; one wave takes 32 patches that
// share P'_k and runs y = x^T P' - m' and gamma = -P' y on the matrix cores (same block skipping);
// the overlap-add is done race-free and in a fixed order by a gather pass.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "jd_common.h"
#include "kernels.h"
#include "jd_adam.h"

namespace jd {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int P = 8;   // patch edge
constexpr int D = 64;  // features per patch
// per component: A fragments [jb 4][st4 4][lane 64][e 4] (P'[pixel 16 st4 + 4 e + (lane >> 4)][16 jb + (lane & 15)])
constexpr int AFRAG_FLOATS = 4 * 4 * 64 * 4;

enum { MODE_MAX = 0, MODE_LSE = 1, MODE_DENSE = 2 };

__host__ __device__ inline unsigned long long best_key(float l, int k) {
  unsigned u = 0;
#if defined(__HIP_DEVICE_COMPILE__)
  u = __float_as_uint(l);
#else
  memcpy(&u, &l, 4);
#endif
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // monotonic map float -> uint
  return ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)k);
}
__device__ inline float best_value(unsigned long long key) {
  unsigned u = (unsigned)(key >> 32);
  u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
  return __uint_as_float(u);
}
__device__ inline int best_component(unsigned long long key) { return (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFu)); }

struct GmmFwdArgs {
  const float* flux;     // (H, W) image  | MODE_DENSE: (n, 64) explicit patches
  const float* afrag;    // K * AFRAG_FLOATS
  const float* mfrag;    // K * 64: [jb 4][g 4][r 4] = -m'[16 jb + 4 g + r]
  const float* const_k;  // K
  int K, H, W, stride, nPx, shift_y, shift_x;
  const int* shift_dev;  // nullable, device [2] = {shift_y, shift_x} residues: read instead of the two members above (use_device_shift)
  int n_begin, n_end;    // linear patch index range (row-major over the patch grid)
  int32_t* argmax_out;   // nullable (MODE_MAX)
  float* value_patch;    // nullable: per patch v | MODE_DENSE: (n, K) out
  double* partials;      // one per block
  const int* run_flag;   // nullable: the kernel returns at once unless *run_flag == run_gen (fallback of the
  int run_gen;           //           screened path, see GmmScreenArgs::flag)
  unsigned long long* best_out;  // nullable (MODE_MAX): per patch (max, arg-max) key, 0 for a filtered patch
};

// v mod n for -n <= v < 2 n: the host normalises the cycle-spin shifts to [0, n), so every coordinate
// (pixel inside the image) - shift is in (-n, n); an integer division here costs ~20 instructions per pixel and
// made the gather the bottleneck of the bucketed kernels.
// The cycle-spin shift of a pass from DEVICE memory (captured hipGraphs replay with the shifts of the step they run for:
// the host uploads them, the launch arguments never change): overwrites the by-value members of the kernel's own copy
// of its arguments.
template <class A>
__device__ __forceinline__ void use_device_shift(A& a) {
  if (a.shift_dev) a.shift_y = a.shift_dev[0], a.shift_x = a.shift_dev[1];
}

__device__ __forceinline__ int wrap(int v, int n) {
  v = v < 0 ? v + n : v;
  return v >= n ? v - n : v;
}

__device__ __forceinline__ float f4_get(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }

// v(lane) + v(lane ^ 16) + v(lane ^ 32) + v(lane ^ 48) on every lane with the gfx950 row / half swaps
// (VALU only; no LDS round trip like ds_bpermute)
__device__ __forceinline__ float sum_lane_groups(float v) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// Patch mean with ONE summation order shared by the forward staging and the backward kernels:
//   S_g = (sum of pixels 4 st + g, st = 0..7 in order) + (the same for st = 8..15),  g = 0..3
//   mean = ((S_0 + S_1) + (S_2 + S_3)) / 64
// y = xbar^T P' is sensitive to the mean at the 1e-4 level (the columns of P' do not sum to zero), so
// the backward kernels must subtract the same bits or the recomputed log-likelihoods (and with them
// the logsumexp responsibilities) would not match the forward pass.
// Backward form: lane group g holds x[st] = pixel 4 st + g of its patch.
__device__ __forceinline__ float patch_mean_groups(const float (&x)[16]) {
  float lo = x[0], hi = x[8];
#pragma unroll
  for (int st = 1; st < 8; ++st) lo += x[st], hi += x[8 + st];
  return sum_lane_groups(lo + hi) * (1.f / 64.f);
}
// Forward staging form: lane half h holds x[s] = pixel 32 h + s, i.e. steps st = 8 h .. 8 h + 7 of every g.
__device__ __forceinline__ float patch_mean_halves(const float (&x)[32]) {
  float t[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float sg = x[g];
#pragma unroll
    for (int k = 1; k < 8; ++k) sg += x[4 * k + g];
    t[g] = sg + __shfl_xor(sg, 32, 64);
  }
  return ((t[0] + t[1]) + (t[2] + t[3])) * (1.f / 64.f);
}

// Fragments of one component held by a lane: A[jb][st4] covers pixel steps 4 st4 .. 4 st4 + 3 of
// coordinate block jb (only st4 <= jb is non-zero for a triangular P), M[jb] the accumulator init.
struct FragBuf {
  float4 a[4][4];
  float4 m[4];
};

template <bool TRI>
__device__ __forceinline__ void load_frags(FragBuf& f, const float4* af, const float4* mf, int k) {
  const float4* ak = af + (size_t)k * (AFRAG_FLOATS / 4);
  const float4* mk = mf + (size_t)k * 16;
#pragma unroll
  for (int jb = 0; jb < 4; ++jb) {
#pragma unroll
    for (int st4 = 0; st4 < 4; ++st4)
      if (!TRI || st4 <= jb) f.a[jb][st4] = ak[(jb * 4 + st4) * 64];
    f.m[jb] = mk[jb * 4];
  }
}

// x[nb * 4 + st4]: B fragments of tile t (two 16-patch halves nb) for pixel steps 4 st4 .. 4 st4 + 3
__device__ __forceinline__ void load_x(float4 (&x)[8], const float* xs_lane, int t) {
#pragma unroll
  for (int q = 0; q < 8; ++q) x[q] = *reinterpret_cast<const float4*>(xs_lane + (t * 8 + q) * 256);
}

// acc[jb][nb] = -m' + sum over the pixel steps of P'^T x  (pixel order = fmaf chain order)
template <bool TRI>
__device__ __forceinline__ void mfma_tile(f32x4 (&acc)[4][2], const FragBuf& f, const float4 (&x)[8]) {
#pragma unroll
  for (int jb = 0; jb < 4; ++jb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) acc[jb][nb] = f32x4{f.m[jb].x, f.m[jb].y, f.m[jb].z, f.m[jb].w};
#pragma unroll
  for (int st = 0; st < 16; ++st) {
#pragma unroll
    for (int jb = TRI ? st / 4 : 0; jb < 4; ++jb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
        acc[jb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4_get(f.a[jb][st >> 2], st & 3),
                                                           f4_get(x[nb * 4 + (st >> 2)], st & 3), acc[jb][nb], 0, 0, 0);
  }
}

// Sum of the 16 squared whitened coordinates a lane holds for 16-patch half nb, on v_pk_fma_f32: fp32
// MFMA and fp32 VALU share the SIMD's FMA lanes (tools/mfma_valu_overlap.hip: every v_fma_f32 beside a
// v_mfma_f32_16x16x4_f32 costs ~5.3 cycles of the wave, a packed one ~6.3 for two fmas), so the epilogue is
// priced per instruction and packing halves its biggest part.  Same summation order in forward and
// backward kernels (the logsumexp responsibilities rely on identical log-likelihoods).
using f32x2 = __attribute__((ext_vector_type(2))) float;

__device__ __forceinline__ float sum_squares(const f32x4 (&acc)[4][2], int nb) {
  f32x2 q = {0.f, 0.f};
#pragma unroll
  for (int jb = 0; jb < 4; ++jb) {
    const f32x2 lo = {acc[jb][nb][0], acc[jb][nb][1]}, hi = {acc[jb][nb][2], acc[jb][nb][3]};
    q = __builtin_elementwise_fma(lo, lo, q);
    q = __builtin_elementwise_fma(hi, hi, q);
  }
  return q[0] + q[1];
}

// Running state of the two 16-patch halves of a tile in LDS: st[nb * 16 + n] = max,
// st[32 + nb * 16 + n] = arg-max | sum-exp.  It is read BEFORE the MFMAs of the stage are issued so that
// the LDS latency is off the critical path of finish_tile.
struct TileState {
  float b[2], s[2];
};

template <int MODE>
__device__ __forceinline__ TileState read_state(const float* st) {
  TileState ts;
  if (MODE != MODE_DENSE) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) ts.b[nb] = st[nb * 16], ts.s[nb] = st[32 + nb * 16];
  }
  return ts;
}

// l = c_k - q / 2 for the two 16-patch halves of a tile, then the branch-free update of the state.
template <int MODE>
__device__ __forceinline__ void finish_tile(const f32x4 (&acc)[4][2], const TileState& ts, float* st, float ck, int k,
                                            const GmmFwdArgs& a, int n_first, bool writer) {
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const float l = fmaf(-0.5f, sum_lane_groups(sum_squares(acc, nb)), ck);  // gmm.py:276-281
    float* s0 = st + nb * 16;
    if (MODE == MODE_MAX) {
      const bool better = l > ts.b[nb];  // strict: the lowest component wins a tie, like torch.max
      s0[0] = better ? l : ts.b[nb];
      s0[32] = better ? __int_as_float(k) : ts.s[nb];
    } else if (MODE == MODE_LSE) {
      const float b = ts.b[nb], sm = ts.s[nb];
      const bool better = l > b;
      const float e = expf(better ? b - l : l - b);
      s0[0] = better ? l : b;
      s0[32] = better ? fmaf(sm, e, 1.f) : sm + e;
    } else {
      const int n = n_first + nb * 16;
      if (writer && n < a.n_end) a.value_patch[(size_t)n * a.K + k] = l;
    }
  }
}

// One component over the block's TB tiles, software pipelined by hand: while the MFMAs of tile t
// issue, the wave has the B operands of tile t+1 in flight from LDS and finishes tile t-1 in the VALU
// shadow of the matrix pipe.  Every stage is one basic block (no branches), TB is even and >= 4.
template <int TB, int MODE, bool TRI>
__device__ __forceinline__ void sweep_tiles(const FragBuf& f, const float* xs_lane, float* st_lane, float ck, int k,
                                            const GmmFwdArgs& a, int n_lane, bool writer) {
  static_assert(TB >= 4 && TB % 2 == 0, "TB must be even and >= 4");
  float4 x0[8], x1[8];
  f32x4 acc0[4][2], acc1[4][2];
  load_x(x0, xs_lane, 0);
  load_x(x1, xs_lane, 1);
  mfma_tile<TRI>(acc0, f, x0);
  for (int t = 1; t < TB - 1; t += 2) {
    const TileState s0 = read_state<MODE>(st_lane + (t - 1) * 64);
    load_x(x0, xs_lane, t + 1);
    mfma_tile<TRI>(acc1, f, x1);
    finish_tile<MODE>(acc0, s0, st_lane + (t - 1) * 64, ck, k, a, n_lane + 32 * (t - 1), writer);
    const TileState s1 = read_state<MODE>(st_lane + t * 64);
    load_x(x1, xs_lane, t + 2);  // t + 2 <= TB - 1
    mfma_tile<TRI>(acc0, f, x0);
    finish_tile<MODE>(acc1, s1, st_lane + t * 64, ck, k, a, n_lane + 32 * t, writer);
  }
  const TileState s0 = read_state<MODE>(st_lane + (TB - 2) * 64);
  const TileState s1 = read_state<MODE>(st_lane + (TB - 1) * 64);
  mfma_tile<TRI>(acc1, f, x1);
  finish_tile<MODE>(acc0, s0, st_lane + (TB - 2) * 64, ck, k, a, n_lane + 32 * (TB - 2), writer);
  finish_tile<MODE>(acc1, s1, st_lane + (TB - 1) * 64, ck, k, a, n_lane + 32 * (TB - 1), writer);
}

// LDS index (in floats) of pixel p of patch c of tile t in B-fragment order:
// [t][nb = c / 16][st4 = p / 16][g = p % 4][n = c % 16][e = (p % 16) / 4]
__device__ __forceinline__ int xs_index(int t, int c, int p) {
  return (((((t * 2 + (c >> 4)) * 4 + (p >> 4)) * 4 + (p & 3)) * 16 + (c & 15)) << 2) + ((p & 15) >> 2);
}

template <int TB, int MODE, bool TRI>
__global__ __launch_bounds__(256, 1) void gmm_fwd_kernel(GmmFwdArgs a) {
  use_device_shift(a);
  if (a.run_flag && *a.run_flag != a.run_gen) return;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xs = lds;                                          // TB * 2048 floats
  float* state = lds + TB * 2048;                           // [4 waves][TB][2][32 patches]
  int* okf = reinterpret_cast<int*>(state + 4 * TB * 64);  // [TB * 32]
  __shared__ double red[4];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile_base = a.n_begin + blockIdx.x * (TB * 32);

  // ---- stage the block's patches (mean subtracted) in MFMA B-operand order ------------------
  {
    const int h = lane >> 5, c = lane & 31;  // lane (h, c) gathers pixels 32 h .. 32 h + 31 of patch c
    for (int t = wave; t < TB; t += 4) {
      const int n = tile_base + 32 * t + c;
      const bool valid = n < a.n_end;
      float x[32];
      bool sel = true;
      if (MODE == MODE_DENSE) {
#pragma unroll
        for (int s = 0; s < 32; ++s) x[s] = valid ? a.flux[(size_t)n * D + 32 * h + s] : 0.f;
      } else {
        const int py = valid ? n / a.nPx : 0, px = valid ? n % a.nPx : 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int yy = wrap(py * a.stride + 4 * h + r - a.shift_y, a.H);
          const float* row = a.flux + (size_t)yy * a.W;
#pragma unroll
          for (int cc = 0; cc < 8; ++cc) {
            const int xx = wrap(px * a.stride + cc - a.shift_x, a.W);
            const float v = valid ? row[xx] : 0.f;
            x[8 * r + cc] = v;
            sel = sel && (v > -1e5f);  // patches/core.py:215
          }
        }
        const float mean = patch_mean_halves(x);  // SubtractMeanPatchNorm, utils/norms.py:100-103
#pragma unroll
        for (int s = 0; s < 32; ++s) x[s] -= mean;
        // NOT `sel && shfl(...)`: the short circuit would keep the lanes with sel == false out of the exchange and the
        // other half of the patch would read a stale register
        const int sel_other = __shfl_xor((int)sel, 32, 64);
        sel = sel && sel_other != 0;
      }
#pragma unroll
      for (int s = 0; s < 32; ++s) xs[xs_index(t, c, 32 * h + s)] = x[s];
      if (h == 0) okf[t * 32 + c] = (valid && sel) ? 1 : 0;
    }
  }
  __syncthreads();

  // ---- this wave's share of the components over all TB tiles ------------------------------------
  const int g = lane >> 4, n16 = lane & 15;
  float* st_lane = state + wave * (TB * 64) + n16;
  if (lane < 32) {
#pragma unroll
    for (int t = 0; t < TB; ++t) state[wave * (TB * 64) + t * 64 + lane] = -INFINITY, state[wave * (TB * 64) + t * 64 + 32 + lane] = 0.f;
  }
  const int k0 = (a.K * wave) / 4, k1 = (a.K * (wave + 1)) / 4;
  const float4* af = reinterpret_cast<const float4*>(a.afrag) + lane;
  const float4* mf = reinterpret_cast<const float4*>(a.mfrag) + g;
  const float* xs_lane = xs + (g * 16 + n16) * 4;
  const int n_lane = tile_base + n16;
  if (k0 < k1) {
    FragBuf f0, f1;
    load_frags<TRI>(f0, af, mf, k0);
    for (int k = k0; k < k1; k += 2) {
      // prefetch is unconditional (clamped): a branch would force a full vmcnt(0) drain
      load_frags<TRI>(f1, af, mf, k + 1 < k1 ? k + 1 : k);
      sweep_tiles<TB, MODE, TRI>(f0, xs_lane, st_lane, a.const_k[k], k, a, n_lane, g == 0);
      load_frags<TRI>(f0, af, mf, k + 2 < k1 ? k + 2 : k);
      if (k + 1 < k1) sweep_tiles<TB, MODE, TRI>(f1, xs_lane, st_lane, a.const_k[k + 1], k + 1, a, n_lane, g == 0);
    }
  }
  if (MODE == MODE_DENSE) return;

  // ---- merge the four component ranges per patch (wave order = component order) ---------------
  __syncthreads();
  double local = 0.0;
  for (int p = threadIdx.x; p < TB * 32; p += 256) {
    const int n = tile_base + p;
    float b = -INFINITY, x1 = 0.f;
    int ar = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float bw = state[w * (TB * 64) + (p >> 5) * 64 + (p & 31)];
      const float sw = state[w * (TB * 64) + (p >> 5) * 64 + 32 + (p & 31)];
      if (MODE == MODE_MAX) {
        if (bw > b) b = bw, ar = __float_as_int(sw);
      } else if (sw > 0.f) {  // online logsumexp merge of (max, sum exp) pairs
        if (bw > b) {
          x1 = x1 * expf(b - bw) + sw;
          b = bw;
        } else {
          x1 += sw * expf(bw - b);
        }
      }
    }
    const float v = MODE == MODE_LSE ? b + logf(x1) : b;
    const bool ok = okf[p] != 0;
    if (n < a.n_end) {
      if (MODE == MODE_MAX && a.argmax_out) a.argmax_out[n] = ok ? ar : -1;
      if (MODE == MODE_MAX && a.best_out) a.best_out[n] = ok ? best_key(b, ar) : 0ull;
      if (a.value_patch) a.value_patch[n] = ok ? v : NAN;
      if (ok) local += (double)v;
    }
  }
  local = wave_sum(local);
  if (lane == 0) red[wave] = local;
  __syncthreads();
  if (threadIdx.x == 0) a.partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ------------------------------------------------------------------------------------------
// Backward, max mode: per patch  gamma = -P'_k* (xbar^T P'_k* - m'_k*),  gbar = gamma - mean(gamma).
// Every patch uses the matrix of ITS arg-max component, so the patches are first bucketed by
// component (counting sort: LDS histograms + one global atomic per bin and block; the order inside a
// bucket does not influence any result); buckets are padded to 32 slots.  One wave then takes a
// 32-slot group, i.e. 32 patches that share P'_k, and runs both products on the matrix cores:
//   Y^T = P'^T Xbar^T - m'      (as in the forward kernel)
//   G^T = P' Y^T                (the Y accumulators ARE the B operand: lane group g holds
//                                Y[16 jb + 4 g + r] in register r of block jb, and the A fragments of
//                                this product are laid out on the host in exactly that order, so no
//                                lane movement / LDS is needed; blocks jb < ib are zero and skipped)
// ------------------------------------------------------------------------------------------
struct GmmBucketArgs {
  const int32_t* argmax;  // global patch index -> component or -1
  int n_begin, n_end, K;
  int* counts;    // K      bin totals (written by the binscan kernel)
  int* offsets;   // K + 1  exclusive scan of the padded counts; offsets[K] = total slots
  int32_t* order; // slot -> global patch index; the slots offsets[k] + counts[k] .. offsets[k + 1] are padding (undefined)
  int32_t* order_n;  // nullable (record sort): slot -> patch of the record, so that the exact kernel needs one hop less
  float* gpatch;  // rows of filtered patches (argmax < 0) are zeroed here (nullable)
  // screened forward pass only (seg_cnt != nullptr): the elements are candidate records in per-wave segments of
  // seg_cap slots of which the first seg_cnt[segment] are used; a record counts only if its upper bound still
  // reaches the final lower bound of its patch
  const int* seg_cnt;
  int seg_cap;
  const int32_t* rec_n;
  const float* rec_ub;
  const float* lfinal;
  // [K][gridDim.x] (bin major: the binscan kernel walks along a bin): per-block bin counts (count kernel), turned
  // into the block's offset inside each bin (binscan)
  int* blk_counts;
  int chunk;    // elements per chunk (multiple of 256): 1024 patches | one record segment (seg_cap)
  int* flag;    // nullable: the scan kernel stores `gen` here (fallback, see GmmScreenArgs) when the padded buckets
  int gen;      //           need more than slot_cap slots
  int slot_cap;
  int* korder;  // nullable: the scan kernel also ranks the bins by size (order of the components for the next screen)
  // logsumexp screen: records count while their upper bound reaches lfinal - margin (0 in max mode), and the scatter
  // kernel also lists every patch's records: ptab[patch * ptab_rows + j] = bucket slot, j < pcount[patch] (more than
  // ptab_rows records of one patch raise the fallback flag)
  float margin;
  int* pcount;
  int32_t* ptab;
  int ptab_rows;
  const int* dense_mark;  // records of marked patches do not count (the dense kernel evaluates those patches)
};

// component of element n, or a negative number if it takes no part (-1: filtered patch)
__device__ __forceinline__ int bucket_key(const GmmBucketArgs& a, int n) {
  if (a.seg_cnt && !(a.rec_ub[n] >= a.lfinal[a.rec_n[n]] - a.margin)) return -2;  // stale record
  if (a.dense_mark && a.dense_mark[a.rec_n[n]] != 0) return -2;
  return a.argmax[n];
}
// number of elements of chunk c that are in use
__device__ __forceinline__ int bucket_chunk_size(const GmmBucketArgs& a, int c) {
  const int left = a.n_end - (a.n_begin + c * a.chunk);
  const int full = left < a.chunk ? left : a.chunk;
  if (!a.seg_cnt) return full;
  const int used = a.seg_cnt[c];  // chunk == record segment
  return used < full ? used : full;
}

// The keys of elements i, i + 256, ... (UN of them; i < size) of a chunk of candidate RECORDS (a.seg_cnt != nullptr) with
// every load unconditional and the independent ones issued together: record -> (patch, bound, component), then the
// patch's final bound.  Through bucket_key, element by element, a thread ran three dependent round trips per record
// (patch, final bound, then -- under the test -- the component).  key = -3: no element (past the chunk's used slots).
template <int UN>
__device__ __forceinline__ void record_keys(const GmmBucketArgs& a, int base, int i, int size, int (&key)[UN], int (&patch)[UN]) {
  int n[UN], kk[UN];
  float ub[UN], lf[UN];
#pragma unroll
  for (int u = 0; u < UN; ++u) n[u] = base + (i + 256 * u < size ? i + 256 * u : i);
#pragma unroll
  for (int u = 0; u < UN; ++u) patch[u] = a.rec_n[n[u]], ub[u] = a.rec_ub[n[u]], kk[u] = a.argmax[n[u]];
#pragma unroll
  for (int u = 0; u < UN; ++u) lf[u] = a.lfinal[patch[u]];
#pragma unroll
  for (int u = 0; u < UN; ++u) {
    bool stale = !(ub[u] >= lf[u] - a.margin);
    if (a.dense_mark) stale = stale || a.dense_mark[patch[u]] != 0;  // (logsumexp screen only)
    key[u] = i + 256 * u < size ? (stale ? -2 : kk[u]) : -3;
  }
}
constexpr int BUCKET_UN = 2;

constexpr int BUCKET_CHUNK = 1024;  // patches per chunk of the backward sort
constexpr int BUCKET_MAX_K = 4096;  // LDS histogram capacity

// Each block walks over chunks blockIdx.x, blockIdx.x + gridDim.x, ... and touches the global counters once per
// bin: with one chunk per block the (bins x blocks) global atomics on a few hundred addresses were the cost.
__global__ __launch_bounds__(256) void gmm_bucket_count_kernel(GmmBucketArgs a) {
  extern __shared__ int hist[];
  for (int k = threadIdx.x; k < a.K; k += 256) hist[k] = 0;
  __syncthreads();
  const int n_chunks = (a.n_end - a.n_begin + a.chunk - 1) / a.chunk;
  for (int c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const int base = a.n_begin + c * a.chunk, size = bucket_chunk_size(a, c);
    if (a.seg_cnt) {  // (uniform) candidate records: batched loads
      for (int i = threadIdx.x; i < size; i += 256 * BUCKET_UN) {
        int key[BUCKET_UN], patch[BUCKET_UN];
        record_keys<BUCKET_UN>(a, base, i, size, key, patch);
#pragma unroll
        for (int u = 0; u < BUCKET_UN; ++u)
          if (key[u] >= 0) atomicAdd(&hist[key[u]], 1);
      }
      continue;
    }
    for (int i = threadIdx.x; i < size; i += 256) {
      const int k = bucket_key(a, base + i);
      if (k >= 0) atomicAdd(&hist[k], 1);
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < a.K; k += 256) a.blk_counts[(size_t)k * gridDim.x + blockIdx.x] = hist[k];
}

// Block k: exclusive prefix over the blocks of bin k's per-block counts (in place) and the bin total.  No global
// atomics anywhere in the sort: with hundreds of blocks hammering a few hundred counters they were its whole cost.
__global__ __launch_bounds__(256) void gmm_bucket_binscan_kernel(GmmBucketArgs a, int n_blk) {
  __shared__ int part[2][256];
  const int k = blockIdx.x;
  const int per = (n_blk + 255) / 256;
  const int b0 = threadIdx.x * per;
  int local = 0;
  int* bin = a.blk_counts + (size_t)k * n_blk;
  for (int b = b0; b < b0 + per && b < n_blk; ++b) local += bin[b];
  int cur = 0;
  part[0][threadIdx.x] = local;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    int v = part[cur][threadIdx.x];
    if ((int)threadIdx.x >= off) v += part[cur][threadIdx.x - off];
    part[cur ^ 1][threadIdx.x] = v;
    cur ^= 1;
    __syncthreads();
  }
  int run = part[cur][threadIdx.x] - local;
  for (int b = b0; b < b0 + per && b < n_blk; ++b) {
    const int v = bin[b];
    bin[b] = run;
    run += v;
  }
  if (threadIdx.x == 255) a.counts[k] = part[cur][255];
}

// The same scan with the thread's counts held in registers (n_blk <= 256 PER): ONE batch of unconditional loads, a wave
// scan by cross-lane moves + the four wave totals through LDS, one batch of stores.  The loop form above runs a load and a
// wait per count, twice (12 dependent round trips at 2040 blocks: 5 us for 1 MB).
template <int PER>
__global__ __launch_bounds__(256) void gmm_bucket_binscan_reg_kernel(GmmBucketArgs a, int n_blk) {
  __shared__ int wave_total[4];
  const int k = blockIdx.x;
  const int per = (n_blk + 255) / 256;
  const int b0 = threadIdx.x * per;
  int* bin = a.blk_counts + (size_t)k * n_blk;
  int vals[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int b = b0 + j;
    const int v = bin[b < n_blk ? b : n_blk - 1];
    vals[j] = (j < per && b < n_blk) ? v : 0;
  }
  int local = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) local += vals[j];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int incl = local;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  if (lane == 63) wave_total[wv] = incl;
  __syncthreads();
  int before = 0;
  for (int w = 0; w < wv; ++w) before += wave_total[w];
  int run = before + incl - local;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int b = b0 + j;
    if (j < per && b < n_blk) bin[b] = run;
    run += vals[j];
  }
  if (threadIdx.x == 255) a.counts[k] = before + incl;
}

static void launch_binscan(const GmmBucketArgs& bk, int K, int n_blk, hipStream_t s) {
  const int per = (n_blk + 255) / 256;
  if (per <= 8)
    gmm_bucket_binscan_reg_kernel<8><<<K, 256, 0, s>>>(bk, n_blk);
  else if (per <= 32)
    gmm_bucket_binscan_reg_kernel<32><<<K, 256, 0, s>>>(bk, n_blk);
  else
    gmm_bucket_binscan_kernel<<<K, 256, 0, s>>>(bk, n_blk);
}

// Exclusive scan of the padded bucket sizes, by EVERY block of the scatter kernel for itself (K bin totals: a few hundred
// loads and one LDS scan -- cheaper than the 6.5 us a dependent single-block launch costs); block 0 also publishes the
// offsets for the kernels that follow, raises the overflow flag and ranks the bins for the next screen.
// off[k] (LDS, K + 1 entries) <- offsets; thread t owns a contiguous segment of bins, the 256 segment sums are scanned
// in LDS (Hillis-Steele).
__device__ __forceinline__ void bucket_offsets(const GmmBucketArgs& a, int* off, int* cnt) {
  __shared__ int part[2][256];
  for (int k = threadIdx.x; k < a.K; k += 256) cnt[k] = a.counts[k];
  __syncthreads();
  const int seg = (a.K + 255) / 256;
  const int k0 = threadIdx.x * seg;
  int local = 0;
  for (int k = k0; k < k0 + seg && k < a.K; ++k) local += (cnt[k] + 31) & ~31;
  int cur = 0;
  part[0][threadIdx.x] = local;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    int v = part[cur][threadIdx.x];
    if ((int)threadIdx.x >= o) v += part[cur][threadIdx.x - o];
    part[cur ^ 1][threadIdx.x] = v;
    cur ^= 1;
    __syncthreads();
  }
  int total = part[cur][threadIdx.x] - local;  // exclusive prefix of this thread's segment
  for (int k = k0; k < k0 + seg && k < a.K; ++k) {
    off[k] = total;
    total += (cnt[k] + 31) & ~31;
  }
  if (threadIdx.x == 255) off[a.K] = part[cur][255];
  __syncthreads();
  if (blockIdx.x != 0) return;
  for (int k = threadIdx.x; k <= a.K; k += 256) a.offsets[k] = off[k];
  if (threadIdx.x == 0 && a.flag && off[a.K] > a.slot_cap) __hip_atomic_store(a.flag, a.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (a.korder) {  // bins by size, largest first (ties: lowest index): the visiting order of the next screen
    for (int k = threadIdx.x; k < a.K; k += 256) {
      const int ck = cnt[k];
      int rank = 0;
      for (int j = 0; j < a.K; ++j) {
        const int cj = cnt[j];
        rank += (cj > ck || (cj == ck && j < k)) ? 1 : 0;
      }
      a.korder[rank] = k;
    }
  }
}

__global__ __launch_bounds__(256) void gmm_bucket_scatter_kernel(GmmBucketArgs a) {
  extern __shared__ int hist[];  // [0, K): the block's next free slot inside each bucket | [K, 2K + 1): offsets | [.., 3K + 1): totals
  int* off = hist + a.K;
  bucket_offsets(a, off, off + a.K + 1);
  const int n_chunks = (a.n_end - a.n_begin + a.chunk - 1) / a.chunk;
  // the block's first slot inside every bucket: bucket offset + the counts of the blocks before it (binscan); the
  // walk over the chunks is the count kernel's, so the numbers match
  for (int k = threadIdx.x; k < a.K; k += 256) hist[k] = off[k] + a.blk_counts[(size_t)k * gridDim.x + blockIdx.x];
  __syncthreads();
  // place the elements (the order inside a bucket does not influence any result)
  for (int c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const int base = a.n_begin + c * a.chunk, size = bucket_chunk_size(a, c);
    if (a.seg_cnt && !a.ptab) {  // (uniform) candidate records of the arg-max screen: batched loads, same walk as the count kernel
      for (int i = threadIdx.x; i < size; i += 256 * BUCKET_UN) {
        int key[BUCKET_UN], patch[BUCKET_UN];
        record_keys<BUCKET_UN>(a, base, i, size, key, patch);
#pragma unroll
        for (int u = 0; u < BUCKET_UN; ++u)
          if (key[u] >= 0) {
            const int pos = atomicAdd(&hist[key[u]], 1);
            a.order[pos] = base + i + 256 * u;
            if (a.order_n) a.order_n[pos] = patch[u];
          }
      }
      continue;
    }
    for (int i = threadIdx.x; i < size; i += 256) {
      const int n = base + i;
      const int k = bucket_key(a, n);
      if (k >= 0) {
        const int pos = atomicAdd(&hist[k], 1);
        a.order[pos] = n;
        if (a.order_n) a.order_n[pos] = a.rec_n[n];
        if (a.ptab) {  // (the order of a patch's entries is whatever the atomics make it: the combine kernel sorts them)
          const int patch = a.rec_n[n];
          const int j = atomicAdd(a.pcount + patch, 1);
          if (j < a.ptab_rows)
            a.ptab[(size_t)patch * a.ptab_rows + j] = pos;
          else
            __hip_atomic_store(a.flag, a.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      } else if (k == -1 && a.gpatch) {  // filtered patch (patches/core.py:215-216): no gradient
        float4* row = reinterpret_cast<float4*>(a.gpatch + (size_t)(n - a.n_begin) * D);
        for (int q = 0; q < D / 4; ++q) row[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
}

// First half of the arg-max backward pass: Y^T = P'^T_k Xbar^T - m'_k for the 2 x 16 patch columns of a wave
// (x[nb][st] = pixel 4 st + g of patch 16 nb + n16, mean subtracted), fragments streamed from L2.
template <bool TRI>
__device__ __forceinline__ void whiten_columns(f32x4 (&y)[4][2], const float (&x)[2][16], const float* afrag,
                                               const float* mfrag, int k, int lane) {
  const float4* ak = reinterpret_cast<const float4*>(afrag) + (size_t)k * (AFRAG_FLOATS / 4) + lane;
  const float4* mk = reinterpret_cast<const float4*>(mfrag) + (size_t)k * 16 + (lane >> 4);
#pragma unroll
  for (int jb = 0; jb < 4; ++jb) {
    const float4 m = mk[jb * 4];
    y[jb][0] = y[jb][1] = f32x4{m.x, m.y, m.z, m.w};
#pragma unroll
    for (int st4 = 0; st4 < 4; ++st4) {
      if (TRI && st4 > jb) continue;
      const float4 A = ak[(jb * 4 + st4) * 64];
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
          y[jb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4_get(A, e), x[nb][4 * st4 + e], y[jb][nb], 0, 0, 0);
    }
  }
}

// Second half of the arg-max backward pass, shared by the bucketed kernel, the fused exact kernel and the fallback:
// G^T = P'_k Y^T (k-step (jb, r) feeds lane group g the value y[jb][nb][r]), gamma = -G, minus its mean over the 64
// pixels (adjoint of the mean subtraction); lane (g, n16) writes pixels 16 ib + 4 g + (0..3) of patch (nb, n16) to
// rows[nb] where valid[nb].  The columns (patches) of the MFMA are independent: zero columns change nothing.
template <bool TRI>
__device__ __forceinline__ void patch_gradient_rows(const f32x4 (&y)[4][2], const float* gfrag, int k, int lane,
                                                    const bool (&valid)[2], float* const (&rows)[2]) {
  const int g = lane >> 4;
  f32x4 gacc[4][2];
  const float4* gk = reinterpret_cast<const float4*>(gfrag) + (size_t)k * (AFRAG_FLOATS / 4) + lane;
#pragma unroll
  for (int ib = 0; ib < 4; ++ib) {
    gacc[ib][0] = gacc[ib][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
      if (TRI && jb < ib) continue;
      const float4 A = gk[(ib * 4 + jb) * 64];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
          gacc[ib][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4_get(A, r), y[jb][nb][r], gacc[ib][nb], 0, 0, 0);
    }
  }
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    float sum = 0.f;
#pragma unroll
    for (int ib = 0; ib < 4; ++ib) sum += (gacc[ib][nb][0] + gacc[ib][nb][1]) + (gacc[ib][nb][2] + gacc[ib][nb][3]);
    const float mean = sum_lane_groups(sum) * (1.f / 64.f);
    if (valid[nb]) {
      float4* out = reinterpret_cast<float4*>(rows[nb]);
#pragma unroll
      for (int ib = 0; ib < 4; ++ib)
        out[4 * ib + g] = make_float4(mean - gacc[ib][nb][0], mean - gacc[ib][nb][1], mean - gacc[ib][nb][2],
                                      mean - gacc[ib][nb][3]);
    }
  }
}

struct GmmBwdArgs {
  const float* flux;
  const float* afrag;  // as in the forward kernel
  const float* mfrag;
  const float* gfrag;  // K * [ib 4][jb 4][lane 64][r 4] = P'[16 ib + (lane & 15)][16 jb + 4 (lane >> 4) + r]
  const int32_t* argmax;
  const int32_t* order;
  const int* offsets;  // offsets[K] = total slots
  const int* counts;   // elements of bucket k: the slots behind them up to offsets[k + 1] are padding
  float* gpatch;       // (n_end - n_begin) * 64
  int K, H, W, stride, nPx, shift_y, shift_x, n_begin, n_end;
  const int* shift_dev;  // nullable, device [2] = {shift_y, shift_x} residues: read instead of the two members above (use_device_shift)
};

template <bool TRI>
__global__ __launch_bounds__(256) void gmm_bwd_max_kernel(GmmBwdArgs a) {
  use_device_shift(a);
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, n16 = lane & 15;
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int n_waves = gridDim.x * 4;
  const int n_groups = a.offsets[a.K] >> 5;
  for (int grp = wave_global; grp < n_groups; grp += n_waves) {
    // slot 0 of a group is always occupied (padding sits at the end of a bucket)
    const int k = __builtin_amdgcn_readfirstlane(a.argmax[__builtin_amdgcn_readfirstlane(a.order[32 * grp])]);
    const int slot_end = __builtin_amdgcn_readfirstlane(a.offsets[k] + a.counts[k]);
    int n[2];
    bool valid[2];
    // ---- B operand: x[nb][st] = pixel 4 st + g of patch 16 nb + n16, mean subtracted ------------
    float x[2][16];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int slot = 32 * grp + 16 * nb + n16;
      valid[nb] = slot < slot_end;
      n[nb] = valid[nb] ? a.order[slot] : -1;
      const int py = valid[nb] ? n[nb] / a.nPx : 0, px = valid[nb] ? n[nb] % a.nPx : 0;
#pragma unroll
      for (int st = 0; st < 16; ++st) {
        const int p = 4 * st + g;  // pixel index: row p / 8, column p % 8
        const int yy = wrap(py * a.stride + (p >> 3) - a.shift_y, a.H);
        const int xx = wrap(px * a.stride + (p & 7) - a.shift_x, a.W);
        x[nb][st] = valid[nb] ? a.flux[(size_t)yy * a.W + xx] : 0.f;
      }
      const float mean = patch_mean_groups(x[nb]);
#pragma unroll
      for (int st = 0; st < 16; ++st) x[nb][st] -= mean;
    }

    f32x4 y[4][2];
    whiten_columns<TRI>(y, x, a.afrag, a.mfrag, k, lane);
    float* rows[2] = {a.gpatch + (size_t)(valid[0] ? n[0] - a.n_begin : 0) * D, a.gpatch + (size_t)(valid[1] ? n[1] - a.n_begin : 0) * D};
    patch_gradient_rows<TRI>(y, a.gfrag, k, lane, valid, rows);
  }
}

// Fallback of the fused backward pass (the screen gave up: *flag == gen, otherwise the kernel returns at once): the
// patches in their natural order, 32 per group; the components of a group differ, so the wave serves one distinct
// component after the other with the other patches' columns zeroed.  Per patch the arithmetic is that of
// gmm_bwd_max_kernel (MFMA columns are independent), i.e. the same bits; slow, but so is the dense forward kernel
// that has just run.  Filtered patches (argmax < 0) get a zero row.
struct GmmBwdFallbackArgs {
  const float* flux;
  const float* afrag;
  const float* mfrag;
  const float* gfrag;
  const int32_t* argmax;  // global patch index -> component or -1
  float* gpatch;          // (n_end - n_begin) * 64
  const int* flag;
  int gen;
  int K, H, W, stride, nPx, shift_y, shift_x, n_begin, n_end;
  const int* shift_dev;  // nullable, device [2] = {shift_y, shift_x} residues: read instead of the two members above (use_device_shift)
};

// the groups grp_begin, grp_begin + grp_step, ... < grp_end of 32 patches (group 0 starts at a.n_begin), one wave each
template <bool TRI>
__device__ __forceinline__ void bwd_fallback_groups(const GmmBwdFallbackArgs& a, int grp_begin, int grp_end, int grp_step) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, n16 = lane & 15;
  for (int grp = grp_begin; grp < grp_end; grp += grp_step) {
    int n[2], kk[2];
    bool pending[2];
    float x[2][16];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int idx = a.n_begin + 32 * grp + 16 * nb + n16;
      const bool in = idx < a.n_end;
      n[nb] = in ? idx : a.n_begin;
      kk[nb] = in ? a.argmax[idx] : -1;
      pending[nb] = kk[nb] >= 0;
      const int py = n[nb] / a.nPx, px = n[nb] % a.nPx;
#pragma unroll
      for (int st = 0; st < 16; ++st) {
        const int p = 4 * st + g;  // pixel index: row p / 8, column p % 8
        const int yy = wrap(py * a.stride + (p >> 3) - a.shift_y, a.H);
        const int xx = wrap(px * a.stride + (p & 7) - a.shift_x, a.W);
        x[nb][st] = pending[nb] ? a.flux[(size_t)yy * a.W + xx] : 0.f;
      }
      const float mean = patch_mean_groups(x[nb]);
#pragma unroll
      for (int st = 0; st < 16; ++st) x[nb][st] -= mean;
      if (in && !pending[nb]) {
        float4* out = reinterpret_cast<float4*>(a.gpatch + (size_t)(idx - a.n_begin) * D);
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) out[4 * ib + g] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    for (;;) {
      const unsigned long long b0 = __ballot(pending[0]), b1 = __ballot(pending[1]);
      if ((b0 | b1) == 0ull) break;
      const int k = __builtin_amdgcn_readfirstlane(b0 ? __shfl(kk[0], __ffsll((long long)b0) - 1) : __shfl(kk[1], __ffsll((long long)b1) - 1));
      bool act[2];
      float xm[2][16];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        act[nb] = pending[nb] && kk[nb] == k;
#pragma unroll
        for (int st = 0; st < 16; ++st) xm[nb][st] = act[nb] ? x[nb][st] : 0.f;
      }
      f32x4 y[4][2];
      whiten_columns<TRI>(y, xm, a.afrag, a.mfrag, k, lane);
      float* rows[2] = {a.gpatch + (size_t)(n[0] - a.n_begin) * D, a.gpatch + (size_t)(n[1] - a.n_begin) * D};
      patch_gradient_rows<TRI>(y, a.gfrag, k, lane, act, rows);
      pending[0] = pending[0] && !act[0];
      pending[1] = pending[1] && !act[1];
    }
  }
}

// ------------------------------------------------------------------------------------------
// Backward, marginalized (logsumexp) mode: d v / d xbar = sum_k r_k gamma_k with the responsibilities
// r_k = exp(l_k - v) (v = logsumexp from the forward pass) and gamma_k = -P'_k y_k.  One wave owns
// GRP groups of 32 patches and walks over ALL components: Y as in the forward kernel, the columns
// of Y scaled by r_k (per patch = per lane), then G += P'_k (r_k Y) accumulated over k in registers.
// Twice the matrix work of the forward pass; fragments are streamed from L2, register double-buffered.
// ------------------------------------------------------------------------------------------
struct GmmBwdLseArgs {
  const float* flux;
  const float* afrag;
  const float* mfrag;
  const float* gfrag;
  const float* const_k;
  double* partials;          // one per block: the sum of the logsumexp values of its patches
  float* gpatch;             // (n_end - n_begin) * 64
  int K, H, W, stride, nPx, shift_y, shift_x, n_begin, n_end;
  const int* shift_dev;  // nullable, device [2] = {shift_y, shift_x} residues: read instead of the two members above (use_device_shift)
  // Behind the logsumexp screen (mark != nullptr): the kernel evaluates the 32-patch groups that hold a marked patch
  // (more candidates than a patch may keep: smooth patches, where most components are within the margin) -- or, after a
  // fallback of the pass (*run_flag == run_gen), all of them -- and leaves v per patch in vpatch (0 for a filtered
  // patch) instead of the partial sums; rows and values of unmarked patches of a visited group are NOT written (the
  // combine kernel owns them).
  const int* run_flag;
  int run_gen;
  const int* mark;
  float* vpatch;
  const int32_t* list;       // the marked patches, compacted (gmm_lse_list_kernel), and their number: a wave works on 64
  const int* list_count;     // of THEM at a time, so that the launch takes as long as their share of the image
};

struct GFrag {
  float4 a[4][4];  // [ib][jb]
};

template <bool TRI>
__device__ __forceinline__ void load_gfrags(GFrag& f, const float4* gf, int k) {
  const float4* gk = gf + (size_t)k * (AFRAG_FLOATS / 4);
#pragma unroll
  for (int ib = 0; ib < 4; ++ib)
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
      if (!TRI || jb >= ib) f.a[ib][jb] = gk[(ib * 4 + jb) * 64];
}

// One component of the logsumexp pass (value AND gradient in one sweep over the components, the way an online softmax
// is accumulated): y = P'^T xbar - m', l = c_k - |y|^2 / 2; the running maximum m of the patch rises to max(m, l), the sum
// S and the gradient accumulator G are rescaled by exp(m_old - m_new) -- a wave-uniform branch, taken only while some
// patch of the wave still sees its maximum rise -- and the component enters with the weight e = exp(l - m):
// S += e, G += P' (e y).  At the end v = m + log S and the gradient row is G / S.
template <bool TRI, int GRP>
__device__ __forceinline__ void lse_component(const FragBuf& f, const GFrag& gfr, float ck, const float4 (&x)[GRP][8],
                                              float (&m)[GRP][2], float (&S)[GRP][2], f32x4 (&G)[GRP][4][2]) {
#pragma unroll
  for (int gi = 0; gi < GRP; ++gi) {
    f32x4 y[4][2];
    mfma_tile<TRI>(y, f, x[gi]);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const float l = fmaf(-0.5f, sum_lane_groups(sum_squares(y, nb)), ck);
      const bool rises = l > m[gi][nb];
      if (__ballot(rises) != 0ull) {
        const float m_new = rises ? l : m[gi][nb];
        const float scale = rises ? expf(m[gi][nb] - m_new) : 1.f;  // (exp(-inf) = 0 the first time: S and G are 0 anyway)
        m[gi][nb] = m_new;
        S[gi][nb] *= scale;
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
          for (int e = 0; e < 4; ++e) G[gi][ib][nb][e] *= scale;
      }
      const float w = expf(l - m[gi][nb]);  // (a NaN l -- a non-finite pixel -- never rises and poisons S: NaN out)
      S[gi][nb] += w;
#pragma unroll
      for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int e = 0; e < 4; ++e) y[jb][nb][e] *= w;
    }
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
      for (int jb = TRI ? ib : 0; jb < 4; ++jb)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
            G[gi][ib][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4_get(gfr.a[ib][jb], e), y[jb][nb][e], G[gi][ib][nb], 0, 0, 0);
  }
}

// Logsumexp value and gradient rows of all patches in ONE pass over the components (dense path of marginalize = True
// with a gradient, and the gated fallback of the screened one): per patch v = logsumexp_k l_k -> one fp64 partial sum
// per block, gamma = -sum_k r_k P'_k y_k minus its mean -> gpatch.  (Until late in round 3 a forward kernel computed v
// first and this kernel evaluated every l_k a second time to form r_k = exp(l_k - v): three matrix products per
// component instead of two.)
template <bool TRI, int GRP>
__global__ __launch_bounds__(256, 1) void gmm_bwd_lse_kernel(GmmBwdLseArgs a) {
  use_device_shift(a);
  const bool everything = !a.mark || *a.run_flag == a.run_gen;
  __shared__ double red[4];
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, n16 = lane & 15;
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int n_waves = gridDim.x * 4;
  const int n_groups = (a.n_end - a.n_begin + 31) / 32;
  const float4* af = reinterpret_cast<const float4*>(a.afrag) + lane;
  const float4* mf = reinterpret_cast<const float4*>(a.mfrag) + g;
  const float4* gf = reinterpret_cast<const float4*>(a.gfrag) + lane;
  double local = 0.0;
  const int n_listed = everything ? 0 : *a.list_count;
  const int n_steps = everything ? n_groups : (n_listed + 31) / 32;
  for (int grp0 = wave_global * GRP; grp0 < n_steps; grp0 += n_waves * GRP) {
    int n[GRP][2];
    bool valid[GRP][2], sel[GRP][2], mine[GRP][2];
    float m[GRP][2], S[GRP][2];
    float4 x[GRP][8];
    f32x4 G[GRP][4][2];
#pragma unroll
    for (int gi = 0; gi < GRP; ++gi)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const int idx = (grp0 + gi) * 32 + nb * 16 + n16;
        if (everything) {
          n[gi][nb] = a.n_begin + idx;
          valid[gi][nb] = n[gi][nb] < a.n_end;
        } else {
          valid[gi][nb] = idx < n_listed;
          n[gi][nb] = valid[gi][nb] ? a.list[idx] : a.n_begin;
        }
        mine[gi][nb] = valid[gi][nb];
      }
#pragma unroll
    for (int gi = 0; gi < GRP; ++gi)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        m[gi][nb] = -INFINITY, S[gi][nb] = 0.f;
        const int py = valid[gi][nb] ? n[gi][nb] / a.nPx : 0, px = valid[gi][nb] ? n[gi][nb] % a.nPx : 0;
        float xv[16];
        int keep = 1;
#pragma unroll
        for (int st = 0; st < 16; ++st) {
          const int p = 4 * st + g;
          const int yy = wrap(py * a.stride + (p >> 3) - a.shift_y, a.H);
          const int xx = wrap(px * a.stride + (p & 7) - a.shift_x, a.W);
          xv[st] = valid[gi][nb] ? a.flux[(size_t)yy * a.W + xx] : 0.f;
          keep &= xv[st] > -1e5f ? 1 : 0;  // patches/core.py:215
        }
        keep &= __shfl_xor(keep, 16, 64);  // the four lane groups hold 16 pixels of the patch each
        keep &= __shfl_xor(keep, 32, 64);
        sel[gi][nb] = valid[gi][nb] && keep != 0;
        const float mean = patch_mean_groups(xv);
#pragma unroll
        for (int st4 = 0; st4 < 4; ++st4)
          x[gi][nb * 4 + st4] = make_float4(xv[4 * st4] - mean, xv[4 * st4 + 1] - mean, xv[4 * st4 + 2] - mean,
                                            xv[4 * st4 + 3] - mean);
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) G[gi][ib][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
      }

    FragBuf f0, f1;
    GFrag g0, g1;
    load_frags<TRI>(f0, af, mf, 0);
    load_gfrags<TRI>(g0, gf, 0);
    for (int k = 0; k < a.K; k += 2) {
      const int kn = k + 1 < a.K ? k + 1 : k;
      load_frags<TRI>(f1, af, mf, kn);
      load_gfrags<TRI>(g1, gf, kn);
      lse_component<TRI, GRP>(f0, g0, a.const_k[k], x, m, S, G);
      const int kn2 = k + 2 < a.K ? k + 2 : k;
      load_frags<TRI>(f0, af, mf, kn2);
      load_gfrags<TRI>(g0, gf, kn2);
      if (k + 1 < a.K) lse_component<TRI, GRP>(f1, g1, a.const_k[k + 1], x, m, S, G);
    }

    // v = m + log S; gamma = -G / S, minus its mean over the 64 pixels (adjoint of the patch-mean subtraction); a
    // filtered patch has no value and no gradient
#pragma unroll
    for (int gi = 0; gi < GRP; ++gi)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const float inv = sel[gi][nb] ? 1.f / S[gi][nb] : 0.f;
        float sum = 0.f;
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
          G[gi][ib][nb] *= inv;
          sum += (G[gi][ib][nb][0] + G[gi][ib][nb][1]) + (G[gi][ib][nb][2] + G[gi][ib][nb][3]);
        }
        const float mean = sum_lane_groups(sum) * (1.f / 64.f);
        if (mine[gi][nb]) {
          float4* out = reinterpret_cast<float4*>(a.gpatch + (size_t)(n[gi][nb] - a.n_begin) * D);
#pragma unroll
          for (int ib = 0; ib < 4; ++ib)
            out[4 * ib + g] = make_float4(mean - G[gi][ib][nb][0], mean - G[gi][ib][nb][1], mean - G[gi][ib][nb][2],
                                          mean - G[gi][ib][nb][3]);
          const float v = sel[gi][nb] ? m[gi][nb] + logf(S[gi][nb]) : 0.f;
          if (g == 0 && a.vpatch) a.vpatch[n[gi][nb]] = v;
          if (g == 0 && sel[gi][nb]) local += (double)v;
        }
      }
  }
  if (a.vpatch) return;  // (the values are summed by gmm_lse_value_kernel)
  local = wave_sum(local);
  if (lane == 0) red[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) a.partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ------------------------------------------------------------------------------------------
// Overlap-add gather: every pixel of the rolled frame sums the contributions of the patches that
// cover it in a fixed order (no float atomics), un-rolls and accumulates into grad.
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// Screened arg-max (max mode, upper triangular precision factors): the same result as gmm_fwd_kernel<MODE_MAX>, bit for bit,
// for a fraction of the fp32 matrix work.
//
//   1. SCREEN (gmm_screen_kernel): every (patch, component) log-likelihood is first evaluated APPROXIMATELY with
//      one fp16 MFMA product, ytilde = fp16(xbar / s_x)^T fp16(P'_k / s_k) (power-of-two scales, fp32 accumulate;
//      v_mfma_f32_32x32x16_f16 runs at
//      16x the rate of the fp32-input MFMA), together with a rigorous bound on its distance to the fp32 value:
//        |ytilde_j - y_j| <= eps |xbar| |P'_k[:, j]|,   eps = 2^-10 + 2^-22 + accumulation  (two fp16 roundings)
//        |ltilde - l|     <= B = sqrt(2 qtilde) e + e^2 / 2 (+ fp32 rounding slack),  e = eps |xbar| |P'_k|_F
//      (Cauchy-Schwarz twice; qtilde = sum_j ytilde_j^2 / 2).  Sweep 1 over the components finds
//      L = max_k (ltilde - B), a lower bound of the true maximum; sweep 2 keeps the components with
//      ltilde + B >= L.  Every other component is provably below the maximum.  Typically 2-5 of 128 survive.
//   2. The surviving (patch, component) pairs are counting-sorted by component (the bucket kernels of the
//      backward pass).
//   3. EXACT (gmm_exact_kernel): groups of 32 pairs that share P'_k are evaluated with the SAME fp32 MFMA chain,
//      mean order and epilogue as gmm_fwd_kernel (bit-identical l), and merged per patch with a 64-bit atomic max
//      on (l, lowest k wins ties) -- order independent, so the result is deterministic.
//   4. gmm_best_kernel decodes (max, arg-max) per patch and sums the values in a fixed order.
// Anything unusual -- a non-finite screening value, more survivors than the per-wave list holds -- raises a
// device flag; the dense fp32 kernel then runs (it is always enqueued and returns at once when the flag is clear)
// and overwrites the per-patch results.  No host synchronisation anywhere.
// ------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int SCREEN_T = 4;        // tiles of 32 patches per wave (2 for small inputs: more waves)
constexpr int SCREEN_CAP = 4096;   // candidate records a wave can hold (128 patches: 32 per patch); multiple of BUCKET_CHUNK
constexpr int A16_BLOCKS = 6;      // non-zero (32 coordinates x 16 pixels) blocks of an upper triangular P'
constexpr float SCREEN_EPS = 0.001f;  // two fp16 roundings 2^-10 + 2^-22, two fp32 accumulations of 64 terms, slack
constexpr int KORDER_MAX_K = 1024;  // the popularity order of the components is maintained up to this K

struct GmmScreenArgs {
  const float* flux;
  const uint4* afrag16;  // K * A16_BLOCKS * 64 lanes * 8 fp16 of P'_k / s_k
  const float* const_k;  // K
  const float* efro_k;   // K: SCREEN_EPS * |P'_k|_F (rounded up)
  const float* sk2_k;    // K: s_k^2, the squared power-of-two scale of the fp16 fragments
  const float* mnorm_k;  // K: 1.001 |m'_k| (0 for a zero-mean component)
  int K, H, W, stride, nPx, shift_y, shift_x, n_begin, n_end;
  const int* shift_dev;  // nullable, device [2] = {shift_y, shift_x} residues: read instead of the two members above (use_device_shift)
  const int* korder;         // K: the order in which the components are visited (most popular first)
  const uint4* xfrag;        // staged patches (gmm_stage_kernel): fp16 B fragments [tile][pixel step][lane],
  const float* xn;           //   1.0001 |xbar|, s_x^2 and validity per [tile * 32 + c]
  const float* xs2;
  const int* ok;
  float* lfinal;             // per patch: max_k (ltilde - B), a lower bound of the true maximum
  int32_t* rec_n;            // [waves][SCREEN_CAP] candidate records: patch (global index),
  int32_t* rec_k;            //                     component,
  float* rec_ub;             //                     upper bound ltilde + B
  int* seg_cnt;              // [waves] records used
  // Fallback flag: a pass that gives up stores its generation number `gen` (> 0, different for consecutive passes
  // of a handle) here; every later kernel of the pass compares the flag with gen.  Nothing ever has to clear it.
  int* flag;
  int gen;
  int* dense_mark;           // logsumexp screen: per patch (global index), zeroed by the staging kernel; set to 1 for a
                             // patch with more candidates than a patch may keep -- its records are dropped and the
                             // dense kernel evaluates it
  // CLOCK instantiation (jd_gmm_screen_clock): block b < clock_cap leaves the shader-clock ticks and the 100 MHz reference
  // ticks between its first and its last instruction at [2 b], [2 b + 1]: the clock the board holds INSIDE this kernel
  unsigned long long* clock_stamps;
  int clock_cap;
};

struct __attribute__((packed, aligned(4))) F4U {  // 16 bytes at a 4-byte aligned address: one global_load_dwordx4
  float x, y, z, w;
};

// Patch staging for the screen (one wave per tile of 32 patches, any number of waves per SIMD): mean-subtracted patches
// as fp16 B fragments in global memory, their norms, scales and validity, and the initial (max, arg-max) keys.  Inside
// the screen kernel -- one wave per SIMD, 512 registers -- this gather was a latency-bound prologue that nothing could
// overlap: 30 us of a 230 us launch at 2048^2.  As a kernel of its own it runs at the memory system's pace; the
// screen then starts with 16 coalesced 16-byte loads per lane.
struct GmmStageArgs {
  const float* flux;
  int H, W, stride, nPx, shift_y, shift_x, n_begin, n_end, n_tiles;
  const int* shift_dev;  // nullable, device [2] = {shift_y, shift_x} residues: read instead of the two members above (use_device_shift)
  uint4* xfrag;              // [tile][pixel step 4][lane 64] = 8 fp16 of xbar / s_x (B fragment of the 32x32x16 MFMA)
  float* xn;                 // [tile * 32 + c] 1.0001 |xbar|
  float* xs2;                // s_x^2
  int* ok;                   // patch takes part (inside the shard, passes the -1e5 filter)
  unsigned long long* best;  // per patch (global index): initialised here
  int* pcount;               // nullable (logsumexp screen): records per patch (global index), zeroed here
  int* dense_mark;           //   and the "evaluate densely" mark of the patch
  int* dense_count;          //   and (one int) the length of the list of marked patches
};

__global__ __launch_bounds__(256) void gmm_stage_kernel(GmmStageArgs a) {
  use_device_shift(a);
  if (a.pcount && blockIdx.x == 0 && threadIdx.x == 0) *a.dense_count = 0;
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= a.n_tiles) return;
  const int h = lane >> 5, c = lane & 31;  // lane (h, c): image rows 2 s + h (pixel step s) of patch c
  const int n = a.n_begin + 32 * tile + c;
  const bool valid = n < a.n_end;
  const int py = valid ? n / a.nPx : 0, px = valid ? n - (n / a.nPx) * a.nPx : 0;
  const int x0 = px * a.stride - a.shift_x;  // in (-W, W)
  const int xb = x0 < 0 ? x0 + a.W : x0;     // first column of the patch in the image, in [0, W)
  const bool straight = xb + 7 < a.W;        // the 8 columns do not wrap around
  float x[32];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const float* row = a.flux + (size_t)wrap(py * a.stride + 2 * s + h - a.shift_y, a.H) * a.W;
    if (straight) {  // two 16-byte loads at a 4-byte aligned address
      const F4U v0 = *reinterpret_cast<const F4U*>(row + xb), v1 = *reinterpret_cast<const F4U*>(row + xb + 4);
      x[8 * s + 0] = v0.x, x[8 * s + 1] = v0.y, x[8 * s + 2] = v0.z, x[8 * s + 3] = v0.w;
      x[8 * s + 4] = v1.x, x[8 * s + 5] = v1.y, x[8 * s + 6] = v1.z, x[8 * s + 7] = v1.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[8 * s + e] = row[wrap(x0 + e, a.W)];
    }
  }
  bool sel = true;
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    x[i] = valid ? x[i] : 0.f;
    sum += x[i];
    sel = sel && (x[i] > -1e5f);  // patches/core.py:215
  }
  const float mean = (sum + __shfl_xor(sum, 32, 64)) * (1.f / 64.f);
  float n2 = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) x[i] -= mean, n2 = fmaf(x[i], x[i], n2);
  n2 += __shfl_xor(n2, 32, 64);
  const int sel_other = __shfl_xor((int)sel, 32, 64);  // unconditional: see gmm_fwd_kernel
  sel = sel && sel_other != 0;
  const bool ok = valid && sel;
  // fp16 operand: xbar / s_x with the power of two s_x that puts max |xbar| into [2^13, 2^14) -- the scaling is
  // exact, nothing overflows (fp16 max 65504), and whatever underflows is below 2^-27 of the largest pixel
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) amax = fmaxf(amax, fabsf(x[i]));
  amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
  int ex = 14;
  if (amax > 0.f && amax < 3.0e38f) (void)frexpf(amax, &ex);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    f16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (_Float16)ldexpf(x[8 * s + e], 14 - ex);
    a.xfrag[((size_t)tile * 4 + s) * 64 + lane] = __builtin_bit_cast(uint4, v);
  }
  if (h == 0) {
    a.xn[tile * 32 + c] = __builtin_sqrtf(n2) * 1.0001f;
    a.xs2[tile * 32 + c] = ldexpf(1.f, 2 * (ex - 14));
    a.ok[tile * 32 + c] = ok ? 1 : 0;
    if (valid) a.best[n] = ok ? best_key(-INFINITY, 0) : 0ull;
    if (valid && a.pcount) a.pcount[n] = 0, a.dense_mark[n] = 0;
  }
}

struct ScreenFrags {
  f16x8 a[A16_BLOCKS];
};

__device__ __forceinline__ void load_frags16(ScreenFrags& f, const uint4* af, int k) {
  const uint4* ak = af + (size_t)k * (A16_BLOCKS * 64);
#pragma unroll
  for (int b = 0; b < A16_BLOCKS; ++b) {
    const uint4 v = ak[b * 64];
    f.a[b] = __builtin_bit_cast(f16x8, v);
  }
}

// ytilde for one tile: coordinate block 0 (j < 32) needs pixel steps 0, 1; block 1 all four
__device__ __forceinline__ void mfma_screen(f32x16 (&acc)[2], const ScreenFrags& f, const f16x8 (&x)[4]) {
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[0], x[0], zero, 0, 0, 0);
  acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[2], x[0], zero, 0, 0, 0);
  acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[1], x[1], acc[0], 0, 0, 0);
  acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[3], x[1], acc[1], 0, 0, 0);
  acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[4], x[2], acc[1], 0, 0, 0);
  acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[5], x[3], acc[1], 0, 0, 0);
}

// The lane's share of q = sum_j ytilde_j^2 (the 32 coordinates of its lane half)
#ifndef JD_SCREEN_SCALAR_SQ
#define JD_SCREEN_SCALAR_SQ 1
#endif
__device__ __forceinline__ float screen_q_half(const f32x16 (&acc)[2]) {
#if JD_SCREEN_SCALAR_SQ
  // two scalar fmaf chains (even / odd registers): the same additions in the same order as the packed form below, but
  // no v_pk_fma_f32 -- beside MFMAs a packed fp32 instruction costs the wave more than the two scalar ones it replaces
  // (MI355X_MICROARCH.md, "price of one filler beside MFMAs"), and hipcc packs only part of them
  float q0 = 0.f, q1 = 0.f;
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      q0 = __builtin_fmaf(acc[b][r], acc[b][r], q0);
      q1 = __builtin_fmaf(acc[b][r + 1], acc[b][r + 1], q1);
    }
  return q0 + q1;
#else
  f32x2 q2 = {0.f, 0.f};
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const f32x2 v = {acc[b][r], acc[b][r + 1]};
      q2 = __builtin_elementwise_fma(v, v, q2);
    }
  return q2[0] + q2[1];
#endif
}

// TWO tiles (A, B) and one component: after the MFMAs lane (h, c) holds half of q for patch c of both tiles.  One
// v_permlane32_swap hands lanes 0-31 both halves of tile A and lanes 32-63 both halves of tile B, so the per-patch
// arithmetic below runs once for the two tiles (per-lane state: half 0 = tile A's patch, half 1 = tile B's):
//   ltilde = ck - q / 2,   |l - ltilde| <= sqrt(q) e + e^2 / 2,  e = eps |xbar| |P'_k|_F + |m'_k|
// (the screen ignores the component mean m'_k: y - m' = ytilde + d with |d| <= eps |xbar| |P'_k|_F + |m'_k|, so a
// mixture with non-zero means only gets wider bounds; the exact stage subtracts the means)
// inflated for the fp32 rounding of q, l, the hardware square root (1 ulp) and of this expression itself:
//   B = sqrt(q) * e1 + 2e-5 q + c2,   e1 = 1.001 e,   c2 = 0.5 e1^2 + 1e-6 |ck| + 1e-30.
// ONE sweep over the components: a component is recorded while its upper bound reaches the running lower bound L of
// the maximum; records made before L rose are dropped later (bucket_key) against the final L.  Visiting the
// components most-popular-first makes L rise early, so few stale records are written.
// The issue slots beside the MFMAs are the budget (about six 4-cycle VALU instructions hide per 32-cycle MFMA).
// LSE (logsumexp screen): a component is recorded while its upper bound reaches L - LSE_MARGIN -- whatever is left out is
// below exp(-25) = 1.4e-11 of the largest term of the sum, 128 components of it below 2e-9 of the sum.
constexpr float LSE_MARGIN = 25.f;
constexpr int LSE_KEEP = 28;  // candidates a patch may keep (a multiple of 4; <= LSE_ROWS, and 128 x (LSE_KEEP + 1) <= SCREEN_CAP)
#ifndef JD_SCREEN_SCHED_NV
#define JD_SCREEN_SCHED_NV 8
#endif
#ifndef JD_SCREEN_LATE_EMIT
#define JD_SCREEN_LATE_EMIT 1  // records of both pairs are written at the END of a component's step (one basic block for
                               // the MFMAs and the squares of a component); 0: inside each pair's epilogue (rounds 1-3)
#endif

// the candidate records of one pair: `mask` = ballot of the candidate lanes
__device__ __forceinline__ void screen_emit(unsigned long long mask, bool cand, float ub, int n, int k, int lane, int& cnt,
                                            int32_t* rec_n, int32_t* rec_k, float* rec_ub, int cap) {
  if (mask) {
    const int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
#if JD_SCREEN_LATE_EMIT
    cand = ((mask >> lane) & 1ull) != 0ull;  // (a lane flag kept alive across the component's step costs it two instructions)
#endif
    if (cand && pos < cap) {  // (the wave's record buffer: uniform base pointers, one 32-bit offset)
      rec_n[pos] = n;
      rec_k[pos] = k;
      rec_ub[pos] = ub;
    }
    cnt += __popcll(mask);
  }
}

template <bool LSE = false>
__device__ __forceinline__ void screen_finish_pair(const f32x16 (&accA)[2], const f32x16 (&accB)[2], float ck, float ack,
                                                   float mnorm, float efro, float xn, float s2, bool ok, float& L,
                                                   float& qacc, int n, int k, int lane, int& cnt, int32_t* rec_n,
                                                   int32_t* rec_k, float* rec_ub, int cap, int& pc, int keep,
                                                   unsigned long long& mask_out, float& ub_out, unsigned long long okmask,
                                                   bool live) {
  const float qa = screen_q_half(accA), qb = screen_q_half(accB);
  const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(qa), __float_as_uint(qb), false, false);
  // lanes 0-31: tile A, lanes 32-63: tile B; s2 = (s_x s_k)^2 undoes the power-of-two operand scales (exactly)
  const float q = (__uint_as_float(sw[0]) + __uint_as_float(sw[1])) * s2;
  qacc += q;  // a NaN / inf anywhere ends up here and raises the fallback flag
  const float e1 = fmaf(efro, xn, mnorm);  // efro carries eps and the factor 1.001; mnorm = 1.001 |m'_k| (see above)
  const float c2 = fmaf(0.5f * e1, e1, ack);
  const float l = fmaf(-0.5f, q, ck);
  const float B = fmaf(__builtin_amdgcn_sqrtf(q), e1, fmaf(2e-5f, q, c2));
  const float ub = l + B;
  // L = max(L, l - B) as ONE v_max_f32 (fmaxf adds a canonicalising v_max in front; a NaN operand loses either way
  // and is caught through qacc)
  unsigned long long mask;
  bool cand;
  if (LSE) {  // a patch keeps at most `keep` candidates; one more marks it for the dense kernel (its records are dropped)
    cand = ok && live && ub >= L - LSE_MARGIN;
    pc += cand ? 1 : 0;
    cand = cand && pc <= keep;
    mask = __ballot(cand);
  } else {
    // the ballot of the compare alone IS its lane mask; `ok` joins as a scalar AND with its own (loop-invariant) ballot --
    // the ballot of `ok && compare` goes through a v_cndmask / v_cmp_ne pair
    mask = __builtin_amdgcn_ballot_w64(ub >= L) & okmask;
    cand = ((mask >> lane) & 1ull) != 0ull;  // (only the early-emit form reads it)
  }
  asm("v_max_f32 %0, %1, %2" : "=v"(L) : "v"(L), "v"(l - B));
#if JD_SCREEN_LATE_EMIT
  mask_out = mask, ub_out = ub;
#else
  screen_emit(mask, cand, ub, n, k, lane, cnt, rec_n, rec_k, rec_ub, cap);
#endif
}

// NP = tile pairs (of 2 x 32 patches) a wave works on.  Two decompositions:
//   KSPLIT = false  every wave owns its NP pairs and walks over ALL components (fragments amortised over 128 patches,
//                   no synchronisation at all): large inputs;
//   KSPLIT = true   the four waves of a block share NP pairs and each takes every fourth component of the visiting
//                   order: a wave's sweep is four times shorter, so a small input (a rank's share of a sharded prior)
//                   still occupies every CU for a short time instead of a few CUs for the full sweep.  Every wave
//                   keeps its own running bound L_w (a valid lower bound of the maximum), the final bound is their
//                   maximum.
constexpr int SCREEN_RB = 512;      // records a wave buffers in LDS before it writes them out (>= 2 x 64)
constexpr int SCREEN_KC_MAX = 512;  // components whose per-component constants are staged in LDS in visiting order

// KC_LDS: (k, c_k, eps |P'_k|_F, s_k^2, |m'_k|) of the component at every position of the visiting order are staged in
// LDS once per block (K <= SCREEN_KC_MAX).  The kernel stores records, so hipcc may not use scalar loads for these
// uniform values; as vector loads from global memory their latency was exposed once per component (a load of
// korder[kk + 1] followed at once by the wait for it).  From LDS they are fetched TWO positions ahead, so that the
// component index is in a register a whole component before the fragment prefetch needs it for its address.
template <int NP, bool KSPLIT, bool KC_LDS, bool LSE = false, bool CLOCK = false>
__global__ __launch_bounds__(256, NP == 1 ? 2 : 1) void gmm_screen_kernel(GmmScreenArgs a) {
  constexpr int NT = 2 * NP;
  unsigned long long clock_t0 = 0, clock_r0 = 0;
  if (CLOCK) clock_t0 = __builtin_amdgcn_s_memtime(), clock_r0 = __builtin_amdgcn_s_memrealtime();
  __shared__ float st_L[KSPLIT ? 4 * NT * 32 : 1];
  // Candidate records are collected in a wave-private LDS buffer and written to the wave's segment in global memory
  // in bulk: a global store inside the sweep is counted by vmcnt like a load, and the compiler -- which cannot know
  // whether the conditional stores were issued -- makes every later wait for the fragment prefetch drain them as well
  // (the waves were parked on s_waitcnt for a fifth of their cycles).
  __shared__ int32_t rb_n[4][SCREEN_RB], rb_k[4][SCREEN_RB];
  __shared__ float rb_ub[4][SCREEN_RB];
  __shared__ int kc_k[KC_LDS ? SCREEN_KC_MAX : 1];
  __shared__ float4 kc_f[KC_LDS ? SCREEN_KC_MAX : 1];
  __shared__ float kc_a[KC_LDS ? SCREEN_KC_MAX : 1];  // 1e-6 |c_k| + 1e-30: the rounding slack of the bound
  if (KC_LDS) {
    for (int i = threadIdx.x; i < a.K; i += 256) {
      const int k = a.korder[i];
      kc_k[i] = k;
      kc_f[i] = make_float4(a.const_k[k], a.efro_k[k], a.sk2_k[k], a.mnorm_k[k]);
      kc_a[i] = fmaf(1e-6f, fabsf(a.const_k[k]), 1e-30f);
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_global = blockIdx.x * 4 + wave;
  const int tile0 = (KSPLIT ? (int)blockIdx.x : wave_global) * NT;  // first of this wave's (block's) NT tiles
  const int base = a.n_begin + tile0 * 32;
  const int h = lane >> 5, c = lane & 31;  // lane (h, c): image rows 2 s + h (pixel step s) of patch c
  float xn[NT], xs2[NT];
  bool ok[NT];
  int nidx[NT];
  f16x8 xf[NT][4];  // the B fragments of the wave's tiles stay in registers for the whole sweep
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int s = 0; s < 4; ++s) xf[t][s] = __builtin_bit_cast(f16x8, a.xfrag[((size_t)(tile0 + t) * 4 + s) * 64 + lane]);
    xn[t] = a.xn[(tile0 + t) * 32 + c], xs2[t] = a.xs2[(tile0 + t) * 32 + c], ok[t] = a.ok[(tile0 + t) * 32 + c] != 0;
    nidx[t] = base + 32 * t + c;
  }
  if (KC_LDS) __syncthreads();  // the constants table
  const uint4* af = a.afrag16 + lane;
  const int seg = __builtin_amdgcn_readfirstlane(wave_global * SCREEN_CAP);
  int32_t* seg_n = a.rec_n + seg;
  int32_t* seg_k = a.rec_k + seg;
  float* seg_ub = a.rec_ub + seg;
  int cnt = 0;    // records already written to the wave's global segment (may exceed SCREEN_CAP: overflow -> fallback)
  int cnt_l = 0;  // records in the LDS buffer
  int32_t* const lb_n = rb_n[wave];
  int32_t* const lb_k = rb_k[wave];
  float* const lb_ub = rb_ub[wave];
  // room for one more emission of up to 64 records?  otherwise write the buffer out (wave-uniform, rare)
  auto flush = [&](bool force) {
    if (!force && cnt_l <= SCREEN_RB - 128) return;  // (room for the two emissions of the next component)
    for (int i = lane; i < cnt_l; i += 64)
      if (cnt + i < SCREEN_CAP) seg_n[cnt + i] = lb_n[i], seg_k[cnt + i] = lb_k[i], seg_ub[cnt + i] = lb_ub[i];
    cnt += cnt_l;
    cnt_l = 0;
  };
  // per-lane state of the two tile pairs: lane half 0 carries the patch of tile 2 p, half 1 that of tile 2 p + 1
  float pxn[NP], pL[NP], pq[NP], ps2[NP];
  bool pok[NP];
  int pn[NP];
  int pc[NP];  // (logsumexp screen) candidates of the lane's patch so far
  unsigned long long okm[NP];  // ballot of pok
  // candidates a patch may keep: 30 x 128 patches fit a wave's record list, and the four waves of a KSPLIT block, which
  // share the patches, stay below the 32 rows of the patch table together
  const int keep = KSPLIT ? LSE_KEEP / 4 : LSE_KEEP;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    ps2[p] = h ? xs2[2 * p + 1] : xs2[2 * p];
    pxn[p] = h ? xn[2 * p + 1] : xn[2 * p];
    pok[p] = h ? ok[2 * p + 1] : ok[2 * p];
    pn[p] = h ? nidx[2 * p + 1] : nidx[2 * p];
    pL[p] = -INFINITY;
    pq[p] = 0.f;
    pc[p] = 0;
    okm[p] = __ballot(pok[p]);
  }

  ScreenFrags f0, f1;
  f32x16 acc[2][2][2];  // [buffer][tile of the pair][coordinate block]: one pair on the matrix pipe, one in the epilogue
  auto issue_pair = [&](f32x16 (&buf)[2][2], const ScreenFrags& f, int p) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      mfma_screen(buf[u], f, xf[2 * p + u]);
    }
  };
  constexpr int KSTEP = KSPLIT ? 4 : 1;
  const int kk0 = KSPLIT ? wave : 0;  // position in the visiting order: wave w takes w, w + 4, ...
  // (component, constants) at a position of the visiting order
  struct KConst {
    int k;
    float ck, ef, sk2, mn, ack;
  };
  auto fetch_consts = [&](int pos) {
    KConst r;
    if (KC_LDS) {
      r.k = kc_k[pos];
      const float4 c4 = kc_f[pos];
      r.ck = c4.x, r.ef = c4.y, r.sk2 = c4.z, r.mn = c4.w, r.ack = kc_a[pos];
    } else {
      r.k = a.korder[pos];
      r.ck = a.const_k[r.k], r.ef = a.efro_k[r.k], r.sk2 = a.sk2_k[r.k], r.mn = a.mnorm_k[r.k];
      r.ack = fmaf(1e-6f, fabsf(r.ck), 1e-30f);
    }
    return r;
  };
  auto clamp_pos = [&](int pos) { return pos < a.K ? pos : (kk0 < a.K ? kk0 : 0); };
  auto fetch_k = [&](int pos) { return KC_LDS ? kc_k[pos] : a.korder[pos]; };
  KConst cur = fetch_consts(clamp_pos(kk0));
  KConst nxt = fetch_consts(clamp_pos(kk0 + KSTEP));  // always one component ahead of `cur` ...
  load_frags16(f0, af, __builtin_amdgcn_readfirstlane(cur.k));
  load_frags16(f1, af, __builtin_amdgcn_readfirstlane(nxt.k));
  // prologue: pair 0 of the first component.  Unconditional (a wave without components computes on the clamped position and
  // drops the result): with the fragments of f0 consumed on EVERY path into the loop the compiler knows them loaded there,
  // and the first half of a component does not wait -- behind a conditional prologue it drained ALL outstanding loads at the
  // top of every other component, the prefetch of the component after next included
  issue_pair(acc[0], f0, 0);
  int k_ahead_next = fetch_k(clamp_pos(kk0 + 2 * KSTEP));
  // One component: `fa` holds its fragments, `fb` those of the next one (requested during the PREVIOUS component).  As
  // soon as the last MFMA that reads `fa` has been issued, the fragments of the component after next are requested into
  // it: 1.75 components (~3000 cycles) ahead of their first use -- with the request at the top of the component that
  // precedes the use the waves were parked on its vmcnt for a fifth of their cycles (SQ_WAIT_ANY).  The loop alternates
  // the two buffers, so no fragment is ever copied; PHASE = parity of the component within this wave's sweep.
  auto component = [&](ScreenFrags& fa, const ScreenFrags& fb, int kk, auto phase, bool live) {
    constexpr int PHASE = decltype(phase)::value;
    const int k = __builtin_amdgcn_readfirstlane(cur.k);
    const float ck = cur.ck, ef = cur.ef, sk2 = cur.sk2, mn = cur.mn;
    const float ack = cur.ack;
    const int k_ahead = k_ahead_next;                    // the component after next: read from LDS a component ago
    k_ahead_next = fetch_k(clamp_pos(kk + 3 * KSTEP));  // (consumed at once it would expose the LDS latency)
    cur = nxt;
    nxt = fetch_consts(clamp_pos(kk + 2 * KSTEP));  // ... and fetched two ahead of its use
    unsigned long long m0 = 0ull, m1 = 0ull;
    float u0 = 0.f, u1 = 0.f;
    if (NP == 2) {
      // pair 1 of k on the matrix pipe while pair 0 of k finishes in its shadow, then pair 0 of k + 1 | pair 1 of k
      issue_pair(acc[1], fa, 1);
      load_frags16(fa, af, __builtin_amdgcn_readfirstlane(k_ahead));  // unconditional (clamped) prefetch
      screen_finish_pair<LSE>(acc[0][0], acc[0][1], ck, ack, mn, ef, pxn[0], ps2[0] * sk2, pok[0], pL[0], pq[0], pn[0], k, lane,
                         cnt_l, lb_n, lb_k, lb_ub, SCREEN_RB, pc[0], keep, m0, u0, live ? okm[0] : 0ull, live);
#if !JD_SCREEN_LATE_EMIT
      flush(false);
#endif
      issue_pair(acc[0], fb, 0);
      screen_finish_pair<LSE>(acc[1][0], acc[1][1], ck, ack, mn, ef, pxn[NP - 1], ps2[NP - 1] * sk2, pok[NP - 1], pL[NP - 1],
                         pq[NP - 1], pn[NP - 1], k, lane, cnt_l, lb_n, lb_k, lb_ub, SCREEN_RB, pc[NP - 1], keep, m1, u1, live ? okm[NP - 1] : 0ull, live);
#if !JD_SCREEN_LATE_EMIT
      flush(false);
#endif
    } else {
      // the only pair of k + 1 on the matrix pipe while the pair of k finishes; the accumulator buffers alternate
      load_frags16(fa, af, __builtin_amdgcn_readfirstlane(k_ahead));  // (fa's MFMAs were issued by the previous component)
      issue_pair(acc[1 - PHASE], fb, 0);
      screen_finish_pair<LSE>(acc[PHASE][0], acc[PHASE][1], ck, ack, mn, ef, pxn[0], ps2[0] * sk2, pok[0], pL[0], pq[0], pn[0], k,
                         lane, cnt_l, lb_n, lb_k, lb_ub, SCREEN_RB, pc[0], keep, m0, u0, live ? okm[0] : 0ull, live);
#if !JD_SCREEN_LATE_EMIT
      flush(false);
#endif
    }
#if JD_SCREEN_SCHED_NV > 0
    // the component's step is one scheduling region: one MFMA, then JD_SCREEN_SCHED_NV vector instructions, 12 NP times --
    // the squares of one pair spread under the MFMAs of the other (left alone, the scheduler bunches the second pair's
    // MFMAs behind its predecessor's epilogue)
    if (NP == 2 && !LSE && !KSPLIT) {
#pragma unroll
      for (int i = 0; i < 12 * NP; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, JD_SCREEN_SCHED_NV, 0);
      }
    }
#endif
#if JD_SCREEN_LATE_EMIT
    // the records of the component, written behind its MFMAs and squares (SCREEN_RB holds two emissions of 64 + the
    // buffered rest: the flush check runs once per component)
    if (m0 | m1) {
      screen_emit(m0, false, u0, pn[0], k, lane, cnt_l, lb_n, lb_k, lb_ub, SCREEN_RB);
      if (NP == 2) screen_emit(m1, false, u1, pn[NP - 1], k, lane, cnt_l, lb_n, lb_k, lb_ub, SCREEN_RB);
      flush(false);
    }
#endif
  };
  // Components go in PAIRS, the loop has one exit: where a wave's share of the components is odd, the second component of
  // its last pair is the (clamped) first position once more with its records suppressed (`live`; its bound changes
  // nothing: L already holds it).  A conditional second component -- or a second exit -- leaves an edge from the end of the
  // first component to the top of the loop, on which that component's prefetch (six loads into f0) is the newest thing in
  // flight: the compiler then makes the first half of EVERY even component wait for all outstanding loads, the prefetch
  // of the component after next included (s_waitcnt vmcnt(5) ... vmcnt(0) at the loop header).
  for (int kk = kk0; kk < a.K; kk += 2 * KSTEP) {
    component(f0, f1, kk, std::integral_constant<int, 0>{}, true);
    component(f1, f0, kk + KSTEP, std::integral_constant<int, 1>{}, kk + KSTEP < a.K);
  }
  bool trouble = false;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    trouble = trouble || (pok[p] && !(pq[p] < 3.0e38f));
    if (LSE && pc[p] > keep && pn[p] < a.n_end) a.dense_mark[pn[p]] = 1;  // (several waves may store the same 1)
    if (KSPLIT)
      st_L[wave * (NT * 32) + (2 * p + h) * 32 + c] = pL[p];
    else if (pn[p] < a.n_end)
      a.lfinal[pn[p]] = pL[p];
  }
  if (KSPLIT) {  // the final lower bound of a patch is the best of the four waves' bounds
    __syncthreads();
    for (int i = threadIdx.x; i < NT * 32; i += 256)
      if (base + i < a.n_end)
        a.lfinal[base + i] = fmaxf(fmaxf(st_L[i], st_L[NT * 32 + i]), fmaxf(st_L[2 * NT * 32 + i], st_L[3 * NT * 32 + i]));
  }
  // (the counting pass of the record sort was tried here, on the wave's own records against the bound it has just
  // computed: the count kernel went away, -7 us, but at one wave per SIMD the re-read of the records is pure latency
  // and the screen grew by 22 us)
  flush(true);
  if (lane == 0) a.seg_cnt[wave_global] = cnt < SCREEN_CAP ? cnt : SCREEN_CAP;
  if (__ballot(trouble) != 0ull || cnt > SCREEN_CAP) {
    if (lane == 0) __hip_atomic_store(a.flag, a.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (CLOCK) {
    __syncthreads();  // every wave of the block is done
    if (threadIdx.x == 0 && (int)blockIdx.x < a.clock_cap) {
      a.clock_stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - clock_t0;
      a.clock_stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - clock_r0;
    }
  }
}

struct GmmExactArgs {
  const float* flux;
  const float* afrag;
  const float* mfrag;
  const float* const_k;
  const int32_t* order_n; // bucket slot -> patch of the record (only the first counts[k] slots of a bucket are written)
  const int* counts;      // K
  const int* offsets;     // K + 1, offsets[K] = total (padded) bucket slots
  const int* flag;
  int gen;                // the pass has fallen back to the dense kernel when *flag == gen
  unsigned long long* best;
  int K, H, W, stride, nPx, shift_y, shift_x;
  const int* shift_dev;  // nullable, device [2] = {shift_y, shift_x} residues: read instead of the two members above (use_device_shift)
  // fused backward pass (grec != nullptr): the gradient row of EVERY surviving record is written to grec[bucket slot]
  // and the key carries the bucket slot instead of the component (slots ascend with the component, so ties still go
  // to the lowest component); gmm_best_kernel turns the winning key into the row the gather kernel reads
  const float* gfrag;
  float* grec;
  float* lrec;  // nullable (logsumexp screen): l of every surviving record by bucket slot, instead of the max merge
#ifdef JD_EXACT_STAMPS  // diagnostic build only (tools/build_variant.sh stamps -DJD_EXACT_STAMPS=1): s_memtime per phase of every group
  unsigned long long* stamps;  // [group][8]
#endif
};

#ifdef JD_EXACT_STAMPS
#define EXACT_STAMP(i)                                                              \
  do {                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                              \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();                     \
    __builtin_amdgcn_sched_barrier(0);                                              \
    if (lane == 0) a.stamps[(size_t)grp * 8 + (i)] = t_;                            \
  } while (0)
#else
#define EXACT_STAMP(i) do {} while (0)
#endif

constexpr int EXACT_PITCH = 68;  // floats per staged patch (64 + pad: 16-byte aligned rows, 2-way bank spread)
#ifndef JD_EXACT_DRAW
#define JD_EXACT_DRAW 1
#endif
constexpr int EXACT_DRAW = JD_EXACT_DRAW;  // groups a wave draws from the work counter at a time
constexpr int EXACT_OFF_LDS = 1025;        // bucket offsets kept in LDS up to K = 1024

// l(n, k) exactly as gmm_fwd_kernel computes it (same mean order, same MFMA chains, same epilogue), for groups of 32
// surviving records that share the component; merged per patch with an order-independent atomic max.  The patches
// of a group are fetched with 16-byte row segments into a wave-private LDS image (the per-pixel gather of the
// backward kernel costs 4x the memory instructions) and read back in B-operand order.
template <bool TRI>
__global__ __launch_bounds__(256) void gmm_exact_kernel(GmmExactArgs a) {
  use_device_shift(a);
  if (*a.flag == a.gen) return;  // the dense kernel takes over
  __shared__ __attribute__((aligned(16))) float stage[4][32 * EXACT_PITCH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, n16 = lane & 15;
  const int wave_global = blockIdx.x * 4 + wave;
  const int n_waves = gridDim.x * 4;
  const int n_groups = a.offsets[a.K] >> 5;
  float* st = stage[wave];
  // Work distribution.  Phase stamps of the groups (diagnostic build -DJD_EXACT_STAMPS, profiles/r03/exact_stamps.txt): a
  // group takes 36 k cycles where its two products need 5 k -- dependent memory round trips at ~2 us each under load --
  // with a q90 / q50 spread of 1.7 in every phase, so with a fixed run of 4 groups per wave the launch lasts as long as
  // its unluckiest wave (85 us against 55 us per wave on average).  A block therefore owns a contiguous run of groups
  // (one or two components: their fragments stay in this CU's L1) and its four waves DRAW them, EXACT_DRAW at a time,
  // from a counter in LDS; the bucket offsets they search sit in LDS too.  (One global counter for all waves was
  // measured at 137-207 us: 3072 returning atomics on one address serialise at ~40 ns each.)  Results do not depend on
  // who evaluates a group (atomicMax merge, gradient rows by bucket slot).
  (void)n_waves, (void)wave_global;
  __shared__ int s_off[EXACT_OFF_LDS];
  __shared__ int s_next;
  const bool off_lds = a.K + 1 <= EXACT_OFF_LDS;
  const int per_block = (n_groups + (int)gridDim.x - 1) / (int)gridDim.x;
  const int b_begin = (int)blockIdx.x * per_block, b_end = b_begin + per_block < n_groups ? b_begin + per_block : n_groups;
  if (threadIdx.x == 0) s_next = b_begin;
  if (off_lds)
    for (int i = threadIdx.x; i <= a.K; i += 256) s_off[i] = a.offsets[i];
  __syncthreads();
  auto offset_of = [&](int kk) { return off_lds ? s_off[kk] : a.offsets[kk]; };
  int k = -1;
  float ck = 0.f;
  float4 A[4][4], M[4];
  for (;;) {
    int g_begin = 0;
    if (lane == 0) g_begin = atomicAdd(&s_next, EXACT_DRAW);
    g_begin = __builtin_amdgcn_readfirstlane(g_begin);
    if (g_begin >= b_end) break;  // (every wave ends here: the counter only grows)
    const int g_end = g_begin + EXACT_DRAW < b_end ? g_begin + EXACT_DRAW : b_end;
  for (int grp = g_begin; grp < g_end; ++grp) {
    EXACT_STAMP(0);  // group start
    int kg = k;
    if (kg < 0 || offset_of(kg) > 32 * grp || offset_of(kg + 1) <= 32 * grp) {
      // the last k with offsets[k] <= 32 grp (buckets are padded to 32: no straddling)
      int lo = 0, hi = a.K;
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (offset_of(mid) <= 32 * grp) lo = mid; else hi = mid;
      }
      kg = lo;
    }
    if (kg != k) {
      k = kg;
      ck = a.const_k[k];
      const float4* ak = reinterpret_cast<const float4*>(a.afrag) + (size_t)k * (AFRAG_FLOATS / 4) + lane;
      const float4* mk = reinterpret_cast<const float4*>(a.mfrag) + (size_t)k * 16 + g;
#pragma unroll
      for (int jb = 0; jb < 4; ++jb) {
        M[jb] = mk[jb * 4];
#pragma unroll
        for (int st4 = 0; st4 < 4; ++st4)
          if (!TRI || st4 <= jb) A[jb][st4] = ak[(jb * 4 + st4) * 64];
      }
    }
    const int nvalid = a.counts[k] - (32 * grp - offset_of(k));  // >= 1
#ifdef JD_EXACT_STAMPS
    if (lane == 0) a.stamps[(size_t)grp * 8 + 7] = (unsigned long long)((nvalid << 8) | (k & 255));
    { float touch = A[0][0].x + M[0].x; asm volatile("" ::"v"(touch)); }  // the fragment loads have arrived
#endif
    EXACT_STAMP(1);  // bucket found, fragments of a new component in registers
    // ---- stage: lane (q = lane / 2, hh = lane % 2) fetches columns 4 hh .. 4 hh + 3 of the 8 rows of record q
    {
      const int q = lane >> 1, hh = lane & 1;
      const bool have = q < nvalid;
      const int n = have ? a.order_n[32 * grp + q] : 0;
      const int py = n / a.nPx, px = n - py * a.nPx;
      const int x0 = px * a.stride + 4 * hh - a.shift_x;  // in (-W, W)
      const bool straight = x0 >= 0 && x0 + 3 < a.W;
      const int xw[4] = {wrap(x0, a.W), wrap(x0 + 1, a.W), wrap(x0 + 2, a.W), wrap(x0 + 3, a.W)};
      float4 rows[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float* row = a.flux + (size_t)wrap(py * a.stride + r - a.shift_y, a.H) * a.W;
        if (straight) {
          const F4U v = *reinterpret_cast<const F4U*>(row + x0);
          rows[r] = make_float4(v.x, v.y, v.z, v.w);
        } else {
          rows[r] = make_float4(row[xw[0]], row[xw[1]], row[xw[2]], row[xw[3]]);
        }
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) *reinterpret_cast<float4*>(st + q * EXACT_PITCH + 8 * r + 4 * hh) = rows[r];
    }
    EXACT_STAMP(2);  // record indices read, patch rows fetched and stored to LDS
    int n[2];
    bool valid[2];
    float x[2][16];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int q = 16 * nb + n16;
      valid[nb] = q < nvalid;
      n[nb] = valid[nb] ? a.order_n[32 * grp + q] : 0;
#pragma unroll
      for (int s4 = 0; s4 < 16; ++s4) x[nb][s4] = valid[nb] ? st[q * EXACT_PITCH + 4 * s4 + g] : 0.f;
      const float mean = patch_mean_groups(x[nb]);
#pragma unroll
      for (int s4 = 0; s4 < 16; ++s4) x[nb][s4] -= mean;
    }
#ifdef JD_EXACT_STAMPS
    { float touch = x[0][0] + x[1][15]; asm volatile("" ::"v"(touch)); }
#endif
    EXACT_STAMP(3);  // patches read back from LDS, means subtracted
    f32x4 y[4][2];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
      y[jb][0] = y[jb][1] = f32x4{M[jb].x, M[jb].y, M[jb].z, M[jb].w};
#pragma unroll
      for (int st4 = 0; st4 < 4; ++st4) {
        if (TRI && st4 > jb) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
            y[jb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4_get(A[jb][st4], e), x[nb][4 * st4 + e], y[jb][nb], 0, 0, 0);
      }
    }
#ifdef JD_EXACT_STAMPS
    { float touch = y[3][1][3] + y[0][0][0]; asm volatile("" ::"v"(touch)); }
#endif
    EXACT_STAMP(4);  // first product (40 x 2 MFMAs) done
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const float l = fmaf(-0.5f, sum_lane_groups(sum_squares(y, nb)), ck);  // = finish_tile of the forward kernel
      const int tie = a.grec ? 32 * grp + 16 * nb + n16 : k;
      if (a.lrec) {
        if (g == 0 && valid[nb]) a.lrec[32 * grp + 16 * nb + n16] = l;
      } else if (g == 0 && valid[nb] && l > -INFINITY) {
        atomicMax(a.best + n[nb], best_key(l, tie));  // NaN never wins (l > b)
      }
    }
    EXACT_STAMP(5);  // value epilogue, atomicMax issued
    if (a.grec) {
      float* rows[2] = {a.grec + (size_t)(32 * grp + n16) * D, a.grec + (size_t)(32 * grp + 16 + n16) * D};
      patch_gradient_rows<TRI>(y, a.gfrag, k, lane, valid, rows);
    }
#ifdef JD_EXACT_STAMPS
    __builtin_amdgcn_s_waitcnt(0);  // the gradient rows have left the wave
#endif
    EXACT_STAMP(6);  // second product + gradient rows stored
  }
  }
}

struct GmmBestArgs {
  const unsigned long long* best;
  int n_begin, n_end;
  int32_t* argmax_out;  // nullable
  double* partials;     // one per block
  // fused backward pass (winner != nullptr): unless the pass fell back (*flag == gen), the low word of a key is the
  // bucket slot of the winning record -> winner[n] (-1: no gradient); the component is looked up only if asked for.
  // After a fallback the keys carry components (dense kernel): they go to argmax_fb for the fallback backward pass of this kernel (fb).
  const int* flag;
  int gen;
  int32_t* winner;
  int32_t* argmax_fb;
  const int32_t* rec_k;
  const int32_t* rec_order;
  // the block that finishes last turns the partial sums into the prior value (what finalize_sum_kernel would do in a
  // launch of its own, same summation order): value_out = [value_out +] scale * sum(partials)
  int* ticket;  // zero between launches
  double scale;
  float* value_out;
  int accumulate;
  // what the NEXT call's host code wants to know, stored into host-mapped memory by the finishing block (no copy, no
  // synchronisation: the host reads whatever pass has landed): {generation, fell back, bucket slots used, patches}
  int* host_stats;          // nullable
  const int* slots_used;    // offsets[K] of the record sort
  // fused backward pass after a fallback (fb.gpatch != nullptr and *flag == gen): every block produces the gradient
  // rows of its own 1024 patches from the components it has just decoded -- the work of a kernel of its own that in the
  // normal case was a 4.6 us launch returning at once
  GmmBwdFallbackArgs fb;
};

constexpr int BEST_CHUNK = 1024;

__global__ __launch_bounds__(256) void gmm_best_kernel(GmmBestArgs a) {
  use_device_shift(a.fb);
  __shared__ double red[4];
  const int base = a.n_begin + blockIdx.x * BEST_CHUNK;
  const bool slots = a.winner && *a.flag != a.gen;
  double local = 0.0;
  // (the block's keys by unconditional loads, all in flight at once: under the bounds test the compiler emitted load, wait,
  // store, wait per 256 patches -- eight dependent round trips in a launch of one block per CU)
  unsigned long long keys[BEST_CHUNK / 256];
#pragma unroll
  for (int i = 0; i < BEST_CHUNK / 256; ++i) {
    const int n = base + i * 256 + threadIdx.x;
    keys[i] = a.best[n < a.n_end ? n : a.n_end - 1];
  }
  if (slots && !a.argmax_out) {  // (block-uniform) the fit's path: nothing but the winner slots to store, no loads in the loop
#pragma unroll
    for (int i = 0; i < BEST_CHUNK / 256; ++i) {
      const int n = base + i * 256 + threadIdx.x;
      if (n < a.n_end) {
        const unsigned long long key = keys[i];
        const bool ok = key != 0ull;
        const float v = best_value(key);
        a.winner[n] = ok && v > -INFINITY ? best_component(key) : -1;
        if (ok) local += (double)v;
      }
    }
  } else
#pragma unroll
  for (int i = 0; i < BEST_CHUNK / 256; ++i) {
    const int n = base + i * 256 + threadIdx.x;
    if (n < a.n_end) {
      const unsigned long long key = keys[i];
      const bool ok = key != 0ull;
      const float v = best_value(key);
      int k = ok ? best_component(key) : -1;
      if (slots) {
        const int slot = ok && v > -INFINITY ? k : -1;  // no record won: component 0 like the plain keys, no gradient
        a.winner[n] = slot;
        if (a.argmax_out) k = ok ? (slot >= 0 ? a.rec_k[a.rec_order[slot]] : 0) : -1;
      } else if (a.argmax_fb) {
        a.argmax_fb[n] = k;
      }
      if (a.argmax_out) a.argmax_out[n] = k;
      if (ok) local += (double)v;
    }
  }
  local = wave_sum(local);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
  __syncthreads();
  if (a.fb.gpatch && !slots && a.winner) {  // (block-uniform: the pass fell back to the dense kernel)
    __syncthreads();                        // this block's argmax_fb entries are written
    const int grp0 = blockIdx.x * (BEST_CHUNK / 32);
    const int n_groups = (a.n_end - a.n_begin + 31) >> 5;
    const int grp1 = grp0 + BEST_CHUNK / 32 < n_groups ? grp0 + BEST_CHUNK / 32 : n_groups;
    bwd_fallback_groups<true>(a.fb, grp0 + (threadIdx.x >> 6), grp1, 4);
  }
  __shared__ int last;
  if (threadIdx.x == 0) {
    a.partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    __threadfence();  // the partial sum is visible device-wide before the ticket is drawn
    last = atomicAdd(a.ticket, 1) == (int)gridDim.x - 1 ? 1 : 0;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  // finalize_sum_kernel's order: thread t adds partials t, t + 256, ..., then the fixed block reduction
  __shared__ double smem[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) {
    const unsigned long long bits = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(a.partials + i),
                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // not through this CU's L1
    acc += __builtin_bit_cast(double, bits);
  }
  const double total = block_sum<256>(acc, smem);
  if (threadIdx.x == 0) {
    double v = a.scale * total;
    if (a.accumulate) v += (double)a.value_out[0];
    a.value_out[0] = (float)v;
    *a.ticket = 0;
    if (a.host_stats) {
      a.host_stats[1] = *a.flag == a.gen ? 1 : 0;
      a.host_stats[2] = *a.slots_used;
      a.host_stats[3] = a.n_end - a.n_begin;
      __threadfence_system();
      a.host_stats[0] = a.gen;  // last: marks the other three as belonging to this pass
    }
  }
}

// ------------------------------------------------------------------------------------------
// Logsumexp mode through the screen (marginalize = True, patches/core.py:242-243): the screen keeps every component whose
// upper bound reaches L - 25 (L = the lower bound of the patch's maximum), the exact kernel evaluates l and the
// gradient row of each surviving (patch, component) record, and this kernel combines the records of a patch:
//   v = m + log sum_j exp(l_j - m),  row = sum_j exp(l_j - m) row_j / sum_j exp(l_j - m)   (m = max_j l_j)
// in ascending order of the bucket slot (= of the component: deterministic, whatever order the scatter kernel's atomics
// listed them in).  What the screen left out is below 2e-9 of the sum.  16 lanes per patch, each with one float4 of the
// 256-byte rows; a block = 16 patches.  After a fallback (*flag == gen) the gated dense kernels have done the work.
struct GmmLseCombineArgs {
  const int* pcount;      // records per patch (global index)
  const int32_t* ptab;    // [patch][rows] bucket slots
  int rows;
  const float* lrec;      // l by bucket slot
  const float* grec;      // gradient rows by bucket slot
  float* gpatch;          // (n_end - n_begin) * 64: the combined rows
  float* vpatch;          // v per patch (global index); 0 for a filtered patch
  const int* mark;        // patches the dense kernel evaluates: not touched here
  int n_begin, n_end;
  const int* flag;
  int gen;
};

constexpr int LSE_ROWS = 32;  // records per patch the patch table holds (more: fallback to the dense kernels)

__global__ __launch_bounds__(256) void gmm_lse_combine_kernel(GmmLseCombineArgs a) {
  __shared__ int s_slot[16][LSE_ROWS];
  __shared__ float s_l[16][LSE_ROWS];
  if (*a.flag == a.gen) return;  // (block-uniform)
  const int grp = threadIdx.x >> 4, part = threadIdx.x & 15;
  const int n = a.n_begin + (int)blockIdx.x * 16 + grp;
  const bool live = n < a.n_end && a.mark[n < a.n_end ? n : a.n_begin] == 0;
  int c = live ? a.pcount[n] : 0;
  if (c > a.rows) c = a.rows;  // (cannot be: the scatter kernel raised the flag)
  // the patch's records, two per lane; rank by bucket slot -> LDS in ascending order
  int slot[2];
  float l[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int j = part + 16 * u;
    slot[u] = j < c ? a.ptab[(size_t)n * a.rows + j] : 0x7fffffff;
    l[u] = j < c ? a.lrec[slot[u]] : -INFINITY;
  }
  int rank[2] = {0, 0};
  for (int j = 0; j < c; ++j) {  // (c is uniform over the 16 lanes of the patch)
    const int other = __shfl(j < 16 ? slot[0] : slot[1], (threadIdx.x & 48) + (j & 15), 64);
    rank[0] += other < slot[0] ? 1 : 0;
    rank[1] += other < slot[1] ? 1 : 0;
  }
#pragma unroll
  for (int u = 0; u < 2; ++u)
    if (part + 16 * u < c) s_slot[grp][rank[u]] = slot[u], s_l[grp][rank[u]] = l[u];
  float m = fmaxf(l[0], l[1]);
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));  // (stays inside the 16 lanes)
  __syncthreads();
  float4 G = make_float4(0.f, 0.f, 0.f, 0.f);
  float S = 0.f;
  for (int r = 0; r < c; ++r) {
    const float e = expf(s_l[grp][r] - m);
    const float4 row = reinterpret_cast<const float4*>(a.grec + (size_t)s_slot[grp][r] * D)[part];
    S += e;
    G.x = fmaf(e, row.x, G.x), G.y = fmaf(e, row.y, G.y), G.z = fmaf(e, row.z, G.z), G.w = fmaf(e, row.w, G.w);
  }
  if (live) {
    const float inv = c > 0 ? 1.f / S : 0.f;  // (no record: a filtered patch -- no value, no gradient)
    reinterpret_cast<float4*>(a.gpatch + (size_t)(n - a.n_begin) * D)[part] = make_float4(G.x * inv, G.y * inv, G.z * inv, G.w * inv);
  }
  if (part == 0 && live) a.vpatch[n] = c > 0 ? m + logf(S) : 0.f;
}

// The marked patches of the pass, compacted: a block ranks the marks of its 1024 patches and reserves its piece of the
// list with one atomicAdd (the order of the pieces is whatever the atomics make it -- every result is stored by patch
// index, so none depends on it)
__global__ __launch_bounds__(256) void gmm_lse_list_kernel(const int* mark, int n_begin, int n_end, const int* flag, int gen,
                                                           int32_t* list, int* count) {
  __shared__ int wave_cnt[4][4];
  __shared__ int base;
  if (*flag == gen) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int first = n_begin + (int)blockIdx.x * 1024;
  bool marked[4];
  int rank[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = first + i * 256 + (int)threadIdx.x;
    marked[i] = n < n_end && mark[n] != 0;
    const unsigned long long b = __ballot(marked[i]);
    rank[i] = __popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[i][wave] = __popcll(b);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int total = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const int c = wave_cnt[i][w];
        wave_cnt[i][w] = total;
        total += c;
      }
    base = total ? atomicAdd(count, total) : 0;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (marked[i]) list[base + wave_cnt[i][wave] + rank[i]] = first + i * 256 + (int)threadIdx.x;
}

// Partial sums of the per-patch values in a fixed order (1024 patches per block, thread t adds patches t, t + 256, ...)
// and the number of patches the dense kernel had to take
__global__ __launch_bounds__(256) void gmm_lse_value_kernel(const float* vpatch, const int* mark, int n_begin, int n_end,
                                                            double* partials, int* marked) {
  __shared__ double red[4];
  __shared__ int redm[4];
  const int base = n_begin + (int)blockIdx.x * 1024;
  double local = 0.0;
  int cnt = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = base + i * 256 + (int)threadIdx.x;
    const bool marked = n < n_end && mark[n] != 0;
    if (n < n_end) local += (double)vpatch[n];
    cnt += __popcll(__ballot(marked));
  }
  local = wave_sum(local);  // (cnt: the wave's marked patches, the same number in every lane)
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local, redm[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    marked[blockIdx.x] = (redm[0] + redm[1]) + (redm[2] + redm[3]);
  }
}

// value_out = [value_out +] scale * sum(partials); also leaves the pass statistics for the host (see
// GmmBestArgs::host_stats): "fell back" = 1 after a fallback, 2 when the dense kernel took more than 60 % of the patches
__global__ __launch_bounds__(256) void gmm_lse_finalize_kernel(const double* partials, const int* marked, int count,
                                                               const int* flag, int gen, double scale, float* value_out,
                                                               int accumulate, int* host_stats, const int* slots_used,
                                                               int patches) {
  __shared__ double smem[4];
  __shared__ int smem_i[4];
  double acc = 0.0;
  int cnt = 0;
  for (int i = threadIdx.x; i < count; i += 256) acc += partials[i], cnt += marked[i];
  const double total = block_sum<256>(acc, smem);
  for (int o = 32; o >= 1; o >>= 1) cnt += __shfl_xor(cnt, o, 64);  // patches the dense kernel took
  if ((threadIdx.x & 63) == 0) smem_i[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    double v = scale * total;
    if (accumulate) v += (double)value_out[0];
    value_out[0] = (float)v;
    if (host_stats) {
      const int n_marked = (smem_i[0] + smem_i[1]) + (smem_i[2] + smem_i[3]);
      // (screen + sort + records cost about a third of a dense pass: beyond 60 % of the patches the dense pass alone is cheaper)
      host_stats[1] = *flag == gen ? 1 : (5 * (long)n_marked > 3 * (long)patches ? 2 : 0);
      host_stats[2] = *slots_used;
      host_stats[3] = patches;
      __threadfence_system();
      host_stats[0] = gen;
    }
  }
}

struct GmmGatherArgs {
  const float* gpatch;
  float* grad;
  int H, W, stride, nPx, nPy, shift_y, shift_x, row_begin, row_end;  // patch-row shard
  const int* shift_dev;  // nullable, device [2] = {shift_y, shift_x} residues: read instead of the two members above (use_device_shift)
  int y_begin, y_end;                                                // rolled-frame pixel rows covered
  float coef;
  // fused backward pass (winner != nullptr and no fallback): the row of patch n is grec[winner[n]] (none if < 0)
  const int32_t* winner;
  const float* grec;
  const int* flag;
  int gen;
  // band output (band != nullptr): instead of accumulating into `grad` at the un-rolled position, the rows
  // [y_begin, y_end) of the ROLLED frame are written (assigned; 0 where no patch of the shard covers a pixel) to
  // band[(Y - y_begin) * W + X] -- the compact piece a rank of a sharded prior exchanges (jd_add_rolled_bands)
  float* band;
  // tile kernel: W % 4 == 0 and 16-byte aligned images: the pixel groups of a thread start at X = shift_x (mod 4), so that
  // their un-rolled column is a multiple of 4 and the gradient image is read and written with 16-byte accesses
  int vec;
  // fused optimizer step (do_step; tile kernel, whole image): instead of grad += coef * sum the kernel forms
  // g = step.grad_flux[pixel] + coef * sum (the other gradient terms, read only) and applies the update of adam_kernel to
  // the pixel -- one pass less over the gradient image and one launch less per step
  int do_step;
  int preload;  // do_step: the step's streams are loaded before the patch rows (JD_GMM_GATHER_PRELOAD=0: behind the barrier)
  AdamArgs step;
};

__global__ __launch_bounds__(256) void gmm_gather_kernel(GmmGatherArgs a) {
#pragma clang fp contract(off)
  use_device_shift(a);
  const int Y = a.y_begin + blockIdx.y;
  const int X = blockIdx.x * 256 + threadIdx.x;
  if (X >= a.W || Y >= a.y_end) return;
  // patch rows py with py*stride <= Y <= py*stride + 7
  int py_hi = Y / a.stride;
  int py_lo = (Y - (P - 1) + a.stride - 1) / a.stride;
  if (Y - (P - 1) < 0) py_lo = 0;
  if (py_lo < a.row_begin) py_lo = a.row_begin;
  if (py_hi > a.row_end - 1) py_hi = a.row_end - 1;
  int px_hi = X / a.stride;
  int px_lo = (X - (P - 1) + a.stride - 1) / a.stride;
  if (X - (P - 1) < 0) px_lo = 0;
  if (px_hi > a.nPx - 1) px_hi = a.nPx - 1;
  const bool slots = a.winner && *a.flag != a.gen;
  float sum = 0.f;
  bool any = false;
  for (int py = py_lo; py <= py_hi; ++py) {
    const int r = Y - py * a.stride;
    for (int px = px_lo; px <= px_hi; ++px) {
      const int cc = X - px * a.stride;
      if (slots) {
        const int slot = a.winner[(size_t)py * a.nPx + px];
        if (slot >= 0) sum += a.grec[(size_t)slot * D + r * P + cc];
      } else {
        const size_t n = (size_t)(py - a.row_begin) * a.nPx + px;
        sum += a.gpatch[n * D + r * P + cc];
      }
      any = true;
    }
  }
  if (a.band) {
    a.band[(size_t)(Y - a.y_begin) * a.W + X] = any ? a.coef * sum : 0.f;
    return;
  }
  if (!any) return;
  const int yy = wrap(Y - a.shift_y, a.H), xx = wrap(X - a.shift_x, a.W);
  a.grad[(size_t)yy * a.W + xx] += a.coef * sum;
}

// The same overlap-add, one 32 x 32 pixel tile of the rolled frame per block (stride >= 4: at most 10 x 10 patches touch a
// tile): the gradient rows of those patches are fetched ONCE, as whole 256-byte rows, into LDS and every pixel sums its
// contributions from there in the order of gmm_gather_kernel (patch rows ascending, then patch columns: the same bits).
// Every gather kernel adds the ROUNDED product coef * sum (`fp contract(off)`: no fused multiply-add -- hipcc's __fmul_rn
// is a plain product that the compiler contracts all the same): the band
// form stores that product and jd_add_rolled_bands adds it later, so a sharded step -- with one rank: RCCL's identity
// collectives -- gives the bits of the un-sharded one (tests/test_gpu_distributed.py).
// The per-pixel kernel reads 4 bytes from each of up to four different rows per thread -- 4x the memory instructions,
// none of them a full line; at 4096^2, where the rows no longer sit in the Infinity Cache, it took 5x the 2048^2 time.
#ifndef JD_GATHER_MAX_P
#define JD_GATHER_MAX_P 9
#endif
// patches per tile and dimension: the tile's first row is a multiple of 32 above y_begin = row_begin * stride, so for
// stride 4 (and 8) it is aligned with the patch grid: 9 rows of patches (4); strides 5, 6, 7 have at most
// floor(38 / s) + 1 = 8, 7, 6.  Columns: the tile's first column is shift_x (mod 4), not aligned with the grid: 10.
constexpr int GATHER_T = 32, GATHER_MAX_P = JD_GATHER_MAX_P, GATHER_MAX_PX = JD_GATHER_MAX_P + 1;

__global__ __launch_bounds__(256) void gmm_gather_tile_kernel(GmmGatherArgs a) {
#pragma clang fp contract(off)
  use_device_shift(a);
  use_device_bias(a.step);
  __shared__ __attribute__((aligned(16))) float rows[GATHER_MAX_P * GATHER_MAX_PX][D];
  const int tid = threadIdx.x;
  const int xoff = a.vec ? ((a.shift_x % 4) + 4) & 3 : 0;
  const int X0 = (int)blockIdx.x * GATHER_T - ((4 - xoff) & 3), Y0 = a.y_begin + blockIdx.y * GATHER_T;
  auto ceil_div_pos = [](int v, int s) { return v <= 0 ? 0 : (v + s - 1) / s; };
  int py0 = ceil_div_pos(Y0 - (P - 1), a.stride), py1 = (Y0 + GATHER_T - 1) / a.stride;
  int px0 = ceil_div_pos(X0 - (P - 1), a.stride), px1 = (X0 + GATHER_T - 1) / a.stride;
  if (py0 < a.row_begin) py0 = a.row_begin;
  if (py1 > a.row_end - 1) py1 = a.row_end - 1;
  if (px1 > a.nPx - 1) px1 = a.nPx - 1;
  int npx = px1 - px0 + 1, npy = py1 - py0 + 1;
  if (npx > GATHER_MAX_PX) npx = GATHER_MAX_PX, px1 = px0 + npx - 1;  // (cannot happen, see above: keeps LDS in bounds)
  if (npy > GATHER_MAX_P) npy = GATHER_MAX_P, py1 = py0 + npy - 1;
  const bool touched = npx > 0 && npy > 0;  // (block-uniform) some patch of the shard touches this tile
  if (!touched && !a.do_step) {
    if (a.band) {
      const int Y = Y0 + (tid >> 3);
      for (int i = 0; i < 4; ++i) {
        const int X = X0 + (tid & 7) * 4 + i;
        if (Y < a.y_end && X >= 0 && X < a.W) a.band[(size_t)(Y - a.y_begin) * a.W + X] = 0.f;
      }
    }
    return;
  }
  // the optimizer step's own streams (gradient, parameter, flux, moments, mask of the thread's four pixels) do not depend
  // on the patch rows: their loads are issued FIRST, so that they are in flight beside the winner -> row chain below
  // instead of behind the block barrier (three dependent round trips per block become two)
  const int Yt = Y0 + (tid >> 3), Xt = X0 + (tid & 7) * 4;
  const bool pre = a.do_step && a.vec && Yt < a.y_end && Xt >= 0 && Xt + 3 < a.W;
  const bool early = pre && a.preload;
  float4 pre_g = make_float4(0.f, 0.f, 0.f, 0.f), pre_t = pre_g, pre_f = pre_g, pre_m = pre_g, pre_v = pre_g;
  float4 pre_k = make_float4(1.f, 1.f, 1.f, 1.f);
  auto load_step_streams = [&]() {
    const AdamArgs& st = a.step;
    const size_t idx = (size_t)wrap(Yt - a.shift_y, a.H) * a.W + wrap(Xt - a.shift_x, a.W);
    pre_g = *reinterpret_cast<const float4*>(st.grad_flux + idx);
    pre_t = *reinterpret_cast<const float4*>(st.theta + idx), pre_f = *reinterpret_cast<const float4*>(st.flux_in + idx);
    if (!st.sgd) pre_m = *reinterpret_cast<const float4*>(st.m + idx), pre_v = *reinterpret_cast<const float4*>(st.v + idx);
    if (st.mask) pre_k = *reinterpret_cast<const float4*>(st.mask + idx);
  };
  if (touched) {
    const bool slots = a.winner && *a.flag != a.gen;
    // a wave's loads return in the order they were issued: FIRST the winner slots of all the thread's patches, then the
    // step's streams, then the rows -- the wait for the slots does not wait for the streams, and the streams have landed
    // by the time the rows have (the step's streams in front of the slots: measured slower, 56.9 against 49.9 us)
    constexpr int PER = (GATHER_MAX_P * GATHER_MAX_PX + 15) / 16;
    int slot_of[PER];  // slots: the winner's bucket slot (< 0: none); else the patch's row of gpatch
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int p = (tid >> 4) + 16 * j;
      slot_of[j] = -1;
      if (p < npy * npx) {
        const int py = py0 + p / npx, px = px0 + p % npx;
        slot_of[j] = slots ? a.winner[(size_t)py * a.nPx + px] : (py - a.row_begin) * a.nPx + px;
      }
    }
    if (early) load_step_streams();
    const float* base = slots ? a.grec : a.gpatch;
    // (every load unconditional -- row 0 stands in for "no row" -- so that all of them are in flight at once: with the load
    // under the condition the compiler emitted load, wait, LDS store per patch, six dependent round trips per thread)
    float4 rowv[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j)
      rowv[j] = reinterpret_cast<const float4*>(base + (size_t)(slot_of[j] >= 0 ? slot_of[j] : 0) * D)[tid & 15];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int p = (tid >> 4) + 16 * j;
      if (p < npy * npx)
        *reinterpret_cast<float4*>(&rows[p][(tid & 15) * 4]) = slot_of[j] >= 0 ? rowv[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  } else if (early) {
    load_step_streams();
  }
  __syncthreads();
  const int Y = Y0 + (tid >> 3);
  if (Y >= a.y_end) return;
  int py_hi = Y / a.stride, py_lo = ceil_div_pos(Y - (P - 1), a.stride);
  if (py_lo < py0) py_lo = py0;
  if (py_hi > py1) py_hi = py1;
  const int yy = wrap(Y - a.shift_y, a.H);
  const int Xg = X0 + (tid & 7) * 4;  // the thread's group of four pixels of the rolled frame
  float sum[4];
  bool any[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int X = Xg + i;
    sum[i] = 0.f, any[i] = false;
    if (X < 0 || X >= a.W || !touched) continue;
    int px_hi = X / a.stride, px_lo = ceil_div_pos(X - (P - 1), a.stride);
    if (px_lo < px0) px_lo = px0;
    if (px_hi > px1) px_hi = px1;
    for (int py = py_lo; py <= py_hi; ++py) {
      const int r = Y - py * a.stride;
      for (int px = px_lo; px <= px_hi; ++px) {
        sum[i] += rows[(py - py0) * npx + (px - px0)][r * P + (X - px * a.stride)];  // (a zero row where a patch has no gradient)
        any[i] = true;
      }
    }
  }
  if (a.band) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (Xg + i >= 0 && Xg + i < a.W) a.band[(size_t)(Y - a.y_begin) * a.W + Xg + i] = any[i] ? a.coef * sum[i] : 0.f;
    return;
  }
  const size_t row = (size_t)yy * a.W;
  if (a.vec && Xg >= 0 && Xg + 3 < a.W) {
    // the un-rolled column of the group is a multiple of 4 and the group does not wrap (W % 4 == 0)
    const size_t idx = row + wrap(Xg - a.shift_x, a.W);
    if (a.do_step) {
      const AdamArgs& st = a.step;  // (`pre` holds here: the loads were issued at the top of the kernel)
      if (!early) load_step_streams();  // (JD_GMM_GATHER_PRELOAD=0: behind the barrier, as before)
      const float4 g4 = pre_g, t4 = pre_t, f4 = pre_f, m4 = pre_m, v4 = pre_v, k4 = pre_k;
      float g[4] = {g4.x, g4.y, g4.z, g4.w};
      float th[4] = {t4.x, t4.y, t4.z, t4.w}, f[4] = {f4.x, f4.y, f4.z, f4.w};
      float m[4] = {m4.x, m4.y, m4.z, m4.w}, v[4] = {v4.x, v4.y, v4.z, v4.w};
      float mk[4] = {k4.x, k4.y, k4.z, k4.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (any[i]) g[i] += a.coef * sum[i];
        adam_pixel(th[i], f[i], m[i], v[i], g[i], mk[i], st);
      }
      *reinterpret_cast<float4*>(st.theta + idx) = make_float4(th[0], th[1], th[2], th[3]);
      *reinterpret_cast<float4*>(st.flux_out + idx) = make_float4(f[0], f[1], f[2], f[3]);
      if (!st.sgd) {
        *reinterpret_cast<float4*>(st.m + idx) = make_float4(m[0], m[1], m[2], m[3]);
        *reinterpret_cast<float4*>(st.v + idx) = make_float4(v[0], v[1], v[2], v[3]);
      }
    } else if (any[0] || any[1] || any[2] || any[3]) {
      float4 g4 = *reinterpret_cast<const float4*>(a.grad + idx);
      if (any[0]) g4.x += a.coef * sum[0];
      if (any[1]) g4.y += a.coef * sum[1];
      if (any[2]) g4.z += a.coef * sum[2];
      if (any[3]) g4.w += a.coef * sum[3];
      *reinterpret_cast<float4*>(a.grad + idx) = g4;
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // a group that straddles the image border (or W % 4 != 0): pixel by pixel
    const int X = Xg + i;
    if (X < 0 || X >= a.W) continue;
    const size_t idx = row + wrap(X - a.shift_x, a.W);
    if (a.do_step) {
      const AdamArgs& st = a.step;
      float g = st.grad_flux[idx];
      if (any[i]) g += a.coef * sum[i];
      float th = st.theta[idx], f = st.flux_in[idx], m = st.sgd ? 0.f : st.m[idx], v = st.sgd ? 0.f : st.v[idx];
      const float mk = st.mask ? st.mask[idx] : 1.f;
      adam_pixel(th, f, m, v, g, mk, st);
      st.theta[idx] = th, st.flux_out[idx] = f;
      if (!st.sgd) st.m[idx] = m, st.v[idx] = v;
    } else if (any[i]) {
      a.grad[idx] += a.coef * sum[i];
    }
  }
}

// grad[un-rolled (Y, X)] += sum over the bands that hold row Y, in band order: the pieces of a sharded prior gradient
// (band b = rows [y_begin[b], y_end[b]) of the rolled frame, at bands + b * chunk) put back into the gradient image.
// Every rank adds the same numbers in the same order: replicas stay bit-identical.
constexpr int BANDS_MAX = 64;
template <int NB>  // band ranges a launch carries: 8, 16 or BANDS_MAX
struct AddBandsArgsT {
  float* grad;
  const float* bands;
  size_t chunk;
  int H, W, shift_y, shift_x, n_bands, y_lo, y_hi;
  const int* shift_dev;  // nullable, device [2] = {shift_y, shift_x} residues: read instead of the two members above (use_device_shift)
  int y_begin[NB], y_end[NB];
};
using AddBandsArgs = AddBandsArgsT<BANDS_MAX>;  // (what the host fills; launches copy the ranges into the size they take)

template <int NB>
static AddBandsArgsT<NB> narrow_bands(const AddBandsArgs& a) {
  AddBandsArgsT<NB> n{};
  n.grad = a.grad, n.bands = a.bands, n.chunk = a.chunk, n.H = a.H, n.W = a.W, n.shift_y = a.shift_y, n.shift_x = a.shift_x;
  n.n_bands = a.n_bands, n.y_lo = a.y_lo, n.y_hi = a.y_hi, n.shift_dev = a.shift_dev;
  for (int b = 0; b < NB; ++b) n.y_begin[b] = a.y_begin[b], n.y_end[b] = a.y_end[b];
  return n;
}

// The loop over the bands is UNROLLED over the NB ranges of the launch: indexing the by-value argument arrays with a runtime
// band number made the compiler copy the whole argument block to scratch in every thread (584 bytes per lane: the band sum +
// optimizer step of a 2048^2 image took 147 us, 46 % of a rank's share of an 8-way step; round 5), and staging the ranges in
// LDS by 64 compile-time compares compiled to 30 000 instructions (204 us).  Launches carry 8, 16 or 64 ranges.
template <int NB>
__global__ __launch_bounds__(256) void add_rolled_bands_kernel(AddBandsArgsT<NB> a) {
  use_device_shift(a);
  const int Y = a.y_lo + blockIdx.y;
  const int X = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (X >= a.W || Y >= a.y_hi) return;
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
  bool any = false;
  const bool vec = (a.W & 3) == 0 && (a.chunk & 3) == 0;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int yb = a.y_begin[b], ye = a.y_end[b];  // (unused ranges are empty: y_begin = y_end = 0)
    if (Y < yb || Y >= ye) continue;
    const float* row = a.bands + (size_t)b * a.chunk + (size_t)(Y - yb) * a.W + X;
    if (vec) {
      const float4 v = *reinterpret_cast<const float4*>(row);
      sum.x += v.x, sum.y += v.y, sum.z += v.z, sum.w += v.w;
    } else {
      sum.x += row[0];
      if (X + 1 < a.W) sum.y += row[1];
      if (X + 2 < a.W) sum.z += row[2];
      if (X + 3 < a.W) sum.w += row[3];
    }
    any = true;
  }
  if (!any) return;
  float* out = a.grad + (size_t)wrap(Y - a.shift_y, a.H) * a.W;
  const float v[4] = {sum.x, sum.y, sum.z, sum.w};
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (X + i < a.W) out[wrap(X + i - a.shift_x, a.W)] += v[i];
}

// The same sum, followed at once by the optimizer step of the pixel (sharded fits: the bands of the prior's gradient are
// its last term): g = grad[pixel] + sum over the bands, the additions of add_rolled_bands_kernel in the same order, then
// adam_pixel -- one pass over the gradient image and one launch less per step.  A thread owns an ALIGNED group of four
// pixels of the un-rolled image (16-byte accesses to the optimizer state; W % 4 == 0) and reads the four rolled-frame
// band values of every band that holds its row one by one.
template <int NB>
__global__ __launch_bounds__(256) void add_rolled_bands_step_kernel(AddBandsArgsT<NB> a, AdamArgs st) {
  use_device_shift(a);
  use_device_bias(st);
  const int yy = blockIdx.y;
  const int xx = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (xx >= a.W) return;
  const int Y = wrap(yy + a.shift_y, a.H);  // rolled-frame row of this image row (shift in [0, H))
  float sum[4] = {0.f, 0.f, 0.f, 0.f};
  bool any = false;
  int X[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) X[i] = wrap(xx + i + a.shift_x, a.W);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int yb = a.y_begin[b], ye = a.y_end[b];
    if (Y < yb || Y >= ye) continue;
    const float* row = a.bands + (size_t)b * a.chunk + (size_t)(Y - yb) * a.W;
#pragma unroll
    for (int i = 0; i < 4; ++i) sum[i] += row[X[i]];
    any = true;
  }
  const size_t idx = (size_t)yy * a.W + xx;
  const float4 g4 = *reinterpret_cast<const float4*>(st.grad_flux + idx);
  float g[4] = {g4.x, g4.y, g4.z, g4.w};
  const float4 t4 = *reinterpret_cast<const float4*>(st.theta + idx), f4 = *reinterpret_cast<const float4*>(st.flux_in + idx);
  float th[4] = {t4.x, t4.y, t4.z, t4.w}, f[4] = {f4.x, f4.y, f4.z, f4.w}, m[4] = {0.f, 0.f, 0.f, 0.f}, v[4] = {0.f, 0.f, 0.f, 0.f};
  float mk[4] = {1.f, 1.f, 1.f, 1.f};
  if (!st.sgd) {
    const float4 m4 = *reinterpret_cast<const float4*>(st.m + idx), v4 = *reinterpret_cast<const float4*>(st.v + idx);
    m[0] = m4.x, m[1] = m4.y, m[2] = m4.z, m[3] = m4.w, v[0] = v4.x, v[1] = v4.y, v[2] = v4.z, v[3] = v4.w;
  }
  if (st.mask) {
    const float4 k4 = *reinterpret_cast<const float4*>(st.mask + idx);
    mk[0] = k4.x, mk[1] = k4.y, mk[2] = k4.z, mk[3] = k4.w;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (any) g[i] += sum[i];
    adam_pixel(th[i], f[i], m[i], v[i], g[i], mk[i], st);
  }
  *reinterpret_cast<float4*>(st.theta + idx) = make_float4(th[0], th[1], th[2], th[3]);
  *reinterpret_cast<float4*>(st.flux_out + idx) = make_float4(f[0], f[1], f[2], f[3]);
  if (!st.sgd) {
    *reinterpret_cast<float4*>(st.m + idx) = make_float4(m[0], m[1], m[2], m[3]);
    *reinterpret_cast<float4*>(st.v + idx) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

}  // namespace jd

// ==========================================================================================
constexpr int SCREEN_CLOCK_CAP = 4096;  // blocks of a screen launch that leave clock stamps

struct GmmPass {  // a pass between its two phases (gmm_prior_impl): what the gather must find unchanged
  bool valid = false;
  int H = 0, W = 0, stride = 0, shift_y = 0, shift_x = 0, row_begin = 0, row_end = 0, marginalize = 0;
  bool fused = false, lse_screened = false;
  int gen = 0;
  const int* shift_dev = nullptr;
};

struct jd_gmm {
  GmmPass pass;
  unsigned long long* clock_stamps = nullptr;  // jd_gmm_screen_clock: 2 x SCREEN_CLOCK_CAP ticks, zero = not written
  int K = 0;
  bool triangular = true;  // every P_k upper triangular -> zero blocks are skipped
  float* afrag = nullptr;
  float* mfrag = nullptr;
  float* const_k = nullptr;
  float* gfrag = nullptr;
  int* bucket = nullptr;  // counts (K) | unused (K) | offsets (K + 1)
  // workspaces (grown on demand)
  int32_t* argmax = nullptr;
  size_t argmax_cap = 0;
  int32_t* order = nullptr;
  size_t order_cap = 0;
  float* gpatch = nullptr;
  size_t gpatch_cap = 0;
  float* vpatch = nullptr;  // logsumexp per patch (marginalized backward)
  size_t vpatch_cap = 0;
  double* partials = nullptr;
  size_t partials_cap = 0;
  int n_cu = 256;
  // screened arg-max (upper triangular mixtures): fp16 fragments, bound constants, work space
  bool screen_ok = false;
  uint4* afrag16 = nullptr;
  float* efro_k = nullptr;
  float* sk2_k = nullptr;
  float* mnorm_k = nullptr;
  unsigned long long* best = nullptr;
  size_t best_cap = 0;
  float* lfinal = nullptr;
  size_t lfinal_cap = 0;
  uint4* xfrag = nullptr;     // staged patches of the screen: fp16 fragments, norms | scales (floats), validity
  size_t xfrag_cap = 0;
  float* xstat = nullptr;
  size_t xstat_cap = 0;
  int* xok = nullptr;
  size_t xok_cap = 0;
  int32_t* rec = nullptr;  // candidate records: patch | component | upper bound (as float), `slots` each
  size_t rec_cap = 0;
  int32_t* rec_order = nullptr;
  size_t rec_order_cap = 0;
  int32_t* rec_order_n = nullptr;  // bucket slot -> patch of the record
  size_t rec_order_n_cap = 0;
  int* seg_cnt = nullptr;
  size_t seg_cnt_cap = 0;
  // logsumexp screen: l per bucket slot, records per patch and their bucket slots, the combine kernel's partial sums
  float* lrec = nullptr;
  size_t lrec_cap = 0;
  int* pcount = nullptr;
  size_t pcount_cap = 0;
  int32_t* ptab = nullptr;
  size_t ptab_cap = 0;
  double* partials_lse = nullptr;
  size_t partials_lse_cap = 0;
  int* dense_mark = nullptr;   // patches the dense kernel evaluates (too many candidates)
  size_t dense_mark_cap = 0;
  int* marked_lse = nullptr;   // their number per block of the value kernel
  size_t marked_lse_cap = 0;
  int32_t* dense_list = nullptr;  // the marked patches, compacted (+ one int in front: their number)
  size_t dense_list_cap = 0;
  // Where (nearly) all components are within the margin of the maximum -- smooth images under a mixture with similar
  // constants -- the logsumexp screen cannot pay: every pass overflows a record list and falls back to the dense
  // kernels after 0.5 ms of screening.  Once a pass has fallen back with the record buffer at its largest, the next
  // lse_skip passes go to the dense kernels directly; then the screen is tried again.
  int lse_skip = 0;
  int lse_seen_gen = 0;
  bool last_pass_lse = false;
  // Gradient rows per patch the record buffer has room for (x 256 B x patches).  Starts at 4; a pass that fell back
  // because it needed more, or filled more than 60 % of it, doubles it for the following passes (up to 32) -- known
  // from the host-mapped statistics the last block of gmm_best_kernel leaves behind, read without synchronisation.
  int rows_per_patch = 4;
  int* host_stats = nullptr;      // hipHostMalloc (mapped): {generation, fell back, bucket slots used, patches}
  int* host_stats_dev = nullptr;  // its device address
  int stats_seen_gen = 0;
  int* blk_counts = nullptr;  // per-block bin counts of the bucket sort
  size_t blk_counts_cap = 0;
  int* korder = nullptr;      // K: visiting order of the components (most survivors in the previous call first)
  int* screen_ctl = nullptr;  // [0] fallback flag (generation stamped) | counts (K) | unused (K) | offsets (K + 1) | ticket
  int gen = 0;                // generation of the current screened pass (1 .. 2^30, never 0)
  // fused backward pass of the screened path
  float* grec = nullptr;      // gradient rows of the surviving records, by bucket slot
  size_t grec_cap = 0;
  int32_t* winner = nullptr;  // patch -> bucket slot of its winning record
  size_t winner_cap = 0;
};

using namespace jd;

template <typename Tp>
static int grow(Tp** ptr, size_t* cap, size_t need) {
  if (need <= *cap) return JD_OK;
  if (*ptr) (void)hipFree(*ptr);
  *ptr = nullptr;
  *cap = 0;
  JD_HIP(hipMalloc(ptr, need * sizeof(Tp)));
  *cap = need;
  return JD_OK;
}

extern "C" int jd_gmm_create(int K, int Dn, const float* prec_chol, const float* mu_prec, const float* const_k,
                             const float* pixel_w, jd_gmm** gmm_out) {
  JD_REQUIRE(gmm_out && prec_chol && mu_prec && const_k && pixel_w, "jd_gmm_create: null argument");
  JD_REQUIRE(K >= 1 && K <= BUCKET_MAX_K, "jd_gmm_create: K = %d out of range [1, %d]", K, BUCKET_MAX_K);
  JD_REQUIRE(Dn == D, "jd_gmm_create: only 8x8 patches (D = 64) are supported, got D = %d", Dn);
  jd_gmm* g = new (std::nothrow) jd_gmm();
  if (!g) return fail(JD_ERR_ALLOC, "jd_gmm_create: out of host memory");
  g->K = K;

  std::vector<float> afrag((size_t)K * AFRAG_FLOATS), gfrag((size_t)K * AFRAG_FLOATS), mfrag((size_t)K * 64),
      prow((size_t)D * D), mrow(D);
  double sw[D];
  for (int j = 0; j < D; ++j) sw[j] = std::sqrt((double)pixel_w[j]);
  bool tri = true;
  for (int k = 0; k < K; ++k) {
    const float* Pk = prec_chol + (size_t)k * D * D;
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < D; ++j) {
        prow[(size_t)i * D + j] = (float)((double)Pk[i * D + j] * sw[j]);  // P'[i][j] = P[i][j] * sqrt(w_j)
        if (i > j && Pk[i * D + j] != 0.f) tri = false;
      }
    for (int j = 0; j < D; ++j) mrow[j] = (float)((double)mu_prec[(size_t)k * D + j] * sw[j]);
    // forward A fragments [jb][st4][lane][e]: P'[pixel 16 st4 + 4 e + (lane >> 4)][16 jb + (lane & 15)]
    for (int jb = 0; jb < 4; ++jb)
      for (int st4 = 0; st4 < 4; ++st4)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 4; ++e) {
            const int pix = 16 * st4 + 4 * e + (lane >> 4), j = 16 * jb + (lane & 15);
            afrag[(size_t)k * AFRAG_FLOATS + (((jb * 4 + st4) * 64 + lane) * 4 + e)] = prow[(size_t)pix * D + j];
          }
    // backward A fragments [ib][jb][lane][r]: P'[16 ib + (lane & 15)][16 jb + 4 (lane >> 4) + r]
    for (int ib = 0; ib < 4; ++ib)
      for (int jb = 0; jb < 4; ++jb)
        for (int lane = 0; lane < 64; ++lane)
          for (int r = 0; r < 4; ++r) {
            const int pix = 16 * ib + (lane & 15), j = 16 * jb + 4 * (lane >> 4) + r;
            gfrag[(size_t)k * AFRAG_FLOATS + (((ib * 4 + jb) * 64 + lane) * 4 + r)] = prow[(size_t)pix * D + j];
          }
    // accumulator init [jb][g][r] = -m'[16 jb + 4 g + r]
    for (int j = 0; j < D; ++j) mfrag[(size_t)k * 64 + j] = -mrow[j];
  }
  g->triangular = tri;
  // screening operands: fp16(P' / s_k) in 32x32x16 A-fragment order, blocks (jb, s) = (0,0) (0,1) (1,0) (1,1) (1,2) (1,3):
  // lane l holds A[row l & 31][k = 8 (l >> 5) + e] = P'[pixel 16 s + 8 (l >> 5) + e][32 jb + (l & 31)]
  bool screenable = true;
  std::vector<uint16_t> a16;
  std::vector<float> efro, sk2, mnorm;
  if (tri) {
    auto to_half = [](float f) -> uint16_t {  // IEEE binary16, round to nearest even (|f| < 65504 here)
      uint32_t u;
      memcpy(&u, &f, 4);
      const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
      const int32_t e = (int32_t)((u >> 23) & 0xFF) - 127;
      uint32_t m = u & 0x7FFFFFu;
      if (e < -25) return sign;               // underflows to zero
      if (e < -14) {                          // subnormal half
        m |= 0x800000u;
        const int shift = -e - 14 + 13;       // 14 .. 24
        uint32_t h = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (h & 1u))) ++h;
        return (uint16_t)(sign | h);
      }
      uint32_t h = ((uint32_t)(e + 15) << 10) | (m >> 13);
      const uint32_t rem = m & 0x1FFFu;
      if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;  // a carry into the exponent is the right answer
      return (uint16_t)(sign | h);
    };
    static const int blk_jb[A16_BLOCKS] = {0, 0, 1, 1, 1, 1}, blk_s[A16_BLOCKS] = {0, 1, 0, 1, 2, 3};
    a16.resize((size_t)K * A16_BLOCKS * 64 * 8);
    efro.resize(K);
    sk2.resize(K);
    mnorm.resize(K);
    for (int k = 0; k < K; ++k) {
      const float* Pk = prec_chol + (size_t)k * D * D;
      double fro = 0.0, amax = 0.0;
      for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) {
          const float v = (float)((double)Pk[i * D + j] * sw[j]);
          fro += (double)v * v;
          amax = std::fmax(amax, std::fabs((double)v));
        }
      // fp16 operand P'_k / s_k with the power of two s_k that puts max |P'_k| into [2^13, 2^14)
      int ex = 14;
      if (amax > 0.0 && std::isfinite(amax)) (void)std::frexp(amax, &ex);
      const double inv_s = std::ldexp(1.0, 14 - ex);
      sk2[k] = (float)std::ldexp(1.0, 2 * (ex - 14));
      efro[k] = (float)(std::sqrt(fro) * (double)SCREEN_EPS * 1.0011);  // includes the 1.001 inflation of the bound
      double m2 = 0.0;
      for (int j = 0; j < D; ++j) {
        const double m = (double)mu_prec[(size_t)k * D + j] * sw[j];
        m2 += m * m;
      }
      mnorm[k] = (float)(std::sqrt(m2) * 1.0011);
      if (!std::isfinite(efro[k]) || !std::isfinite(mnorm[k]) || !(sk2[k] > 0.f) || !std::isfinite(sk2[k])) screenable = false;
      for (int b = 0; b < A16_BLOCKS; ++b)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 8; ++e) {
            const int pix = 16 * blk_s[b] + 8 * (lane >> 5) + e, j = 32 * blk_jb[b] + (lane & 31);
            const float v = (float)((double)Pk[pix * D + j] * sw[j] * inv_s);
            a16[(((size_t)k * A16_BLOCKS + b) * 64 + lane) * 8 + e] = to_half(v);
          }
    }
    if (!screenable) a16.clear();
  }
  auto upload = [&](float** dst, const float* src, size_t n) -> int {
    JD_HIP(hipMalloc(dst, n * sizeof(float)));
    JD_HIP(hipMemcpy(*dst, src, n * sizeof(float), hipMemcpyHostToDevice));
    return JD_OK;
  };
  int rc;
  if ((rc = upload(&g->afrag, afrag.data(), afrag.size())) || (rc = upload(&g->mfrag, mfrag.data(), mfrag.size())) ||
      (rc = upload(&g->const_k, const_k, K)) || (rc = upload(&g->gfrag, gfrag.data(), gfrag.size()))) {
    jd_gmm_destroy(g);
    return rc;
  }
  if (hipMalloc(&g->bucket, (size_t)(3 * K + 1) * sizeof(int)) != hipSuccess ||
      hipMemset(g->bucket, 0, (size_t)(3 * K + 1) * sizeof(int)) != hipSuccess) {
    jd_gmm_destroy(g);
    return fail(JD_ERR_ALLOC, "jd_gmm_create: hipMalloc of the bucket counters failed");
  }
  if (!a16.empty()) {
    if (hipMalloc(&g->afrag16, a16.size() * sizeof(uint16_t)) != hipSuccess ||
        hipMemcpy(g->afrag16, a16.data(), a16.size() * sizeof(uint16_t), hipMemcpyHostToDevice) != hipSuccess ||
        (rc = upload(&g->efro_k, efro.data(), efro.size())) || (rc = upload(&g->sk2_k, sk2.data(), sk2.size())) ||
        (rc = upload(&g->mnorm_k, mnorm.data(), mnorm.size())) ||
        hipMalloc(&g->screen_ctl, (size_t)(3 * K + 3) * sizeof(int)) != hipSuccess ||
        hipMemset(g->screen_ctl, 0, (size_t)(3 * K + 3) * sizeof(int)) != hipSuccess ||
        hipMalloc(&g->korder, (size_t)K * sizeof(int)) != hipSuccess) {
      jd_gmm_destroy(g);
      return fail(JD_ERR_ALLOC, "jd_gmm_create: allocation of the screening operands failed");
    }
    std::vector<int> identity(K);
    for (int k = 0; k < K; ++k) identity[k] = k;
    if (hipMemcpy(g->korder, identity.data(), (size_t)K * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
      jd_gmm_destroy(g);
      return fail(JD_ERR_HIP, "jd_gmm_create: upload of the component order failed");
    }
    g->screen_ok = true;
    // statistics of the last finished pass in host-mapped memory (optional: without it the record buffer keeps its
    // initial capacity)
    void* mapped = nullptr;
    if (!opt_is_set(OPT_GMM_NO_HOST_STATS) &&
        hipHostMalloc(reinterpret_cast<void**>(&g->host_stats), 4 * sizeof(int), hipHostMallocMapped) == hipSuccess) {
      memset(g->host_stats, 0, 4 * sizeof(int));
      if (hipHostGetDevicePointer(&mapped, g->host_stats, 0) == hipSuccess) {
        g->host_stats_dev = static_cast<int*>(mapped);
      } else {
        (void)hipHostFree(g->host_stats);
        g->host_stats = nullptr;
      }
    } else {
      g->host_stats = nullptr;
    }
  }
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
    g->n_cu = prop.multiProcessorCount;
  *gmm_out = g;
  return JD_OK;
}

extern "C" int jd_gmm_destroy(jd_gmm* g) {
  if (!g) return JD_OK;
  (void)hipDeviceSynchronize();
  if (g->clock_stamps) (void)hipFree(g->clock_stamps);
  for (float* p : {g->afrag, g->mfrag, g->const_k, g->gfrag, g->gpatch, g->vpatch})
    if (p) (void)hipFree(p);
  if (g->argmax) (void)hipFree(g->argmax);
  if (g->afrag16) (void)hipFree(g->afrag16);
  if (g->efro_k) (void)hipFree(g->efro_k);
  if (g->sk2_k) (void)hipFree(g->sk2_k);
  if (g->mnorm_k) (void)hipFree(g->mnorm_k);
  if (g->best) (void)hipFree(g->best);
  if (g->lfinal) (void)hipFree(g->lfinal);
  if (g->xfrag) (void)hipFree(g->xfrag);
  if (g->lrec) (void)hipFree(g->lrec);
  if (g->pcount) (void)hipFree(g->pcount);
  if (g->ptab) (void)hipFree(g->ptab);
  if (g->partials_lse) (void)hipFree(g->partials_lse);
  if (g->dense_mark) (void)hipFree(g->dense_mark);
  if (g->marked_lse) (void)hipFree(g->marked_lse);
  if (g->dense_list) (void)hipFree(g->dense_list);
  if (g->xstat) (void)hipFree(g->xstat);
  if (g->xok) (void)hipFree(g->xok);
  if (g->rec) (void)hipFree(g->rec);
  if (g->rec_order) (void)hipFree(g->rec_order);
  if (g->rec_order_n) (void)hipFree(g->rec_order_n);
  if (g->seg_cnt) (void)hipFree(g->seg_cnt);
  if (g->host_stats) (void)hipHostFree(g->host_stats);
  if (g->korder) (void)hipFree(g->korder);
  if (g->blk_counts) (void)hipFree(g->blk_counts);
  if (g->grec) (void)hipFree(g->grec);
  if (g->winner) (void)hipFree(g->winner);
  if (g->screen_ctl) (void)hipFree(g->screen_ctl);
  if (g->order) (void)hipFree(g->order);
  if (g->bucket) (void)hipFree(g->bucket);
  if (g->partials) (void)hipFree(g->partials);
  delete g;
  return JD_OK;
}

extern "C" int jd_gmm_is_triangular(const jd_gmm* g) { return g ? (g->triangular ? 1 : 0) : -1; }

// Tiles per block: the choice that minimises (rounds over the CUs) x (tiles per block); ties go to
// the larger block (fewer fragment re-reads).
static int pick_block_tiles(long n_patches, int n_cu) {
  {  // tuning override
    const int t = opt_value(OPT_GMM_BLOCK_TILES, 0);
    if (t == 4 || t == 8 || t == 16) return t;
  }
  const long nt = (n_patches + 31) / 32;
  int best_tb = 16;
  long best_cost = -1;
  for (int tb : {16, 8, 4}) {
    const long blocks = (nt + tb - 1) / tb;
    const long cost = ((blocks + n_cu - 1) / n_cu) * tb;
    if (best_cost < 0 || cost < best_cost) best_cost = cost, best_tb = tb;
  }
  return best_tb;
}

template <int TB, int MODE, bool TRI>
static int launch_fwd_tb(const GmmFwdArgs& a, unsigned blocks, hipStream_t s) {
  const size_t lds = (size_t)(TB * 2048 + 4 * TB * 64 + TB * 32) * sizeof(float);
  static bool configured = false;
  if (!configured) {
    JD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gmm_fwd_kernel<TB, MODE, TRI>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    configured = true;
  }
  gmm_fwd_kernel<TB, MODE, TRI><<<blocks, 256, lds, s>>>(a);
  JD_LAUNCH_CHECK();
  return JD_OK;
}

// writes one fp64 partial sum per block; *n_partials = number of blocks
template <int MODE>
static int launch_fwd(const GmmFwdArgs& a, bool tri, int n_cu, hipStream_t s, int* n_partials) {
  const long n = a.n_end - a.n_begin;
  if (opt_is_set(OPT_GMM_DENSE)) tri = false;  // tuning / testing: force the dense variant
  const int tb = pick_block_tiles(n, n_cu);
  const unsigned blocks = (unsigned)((n + 32L * tb - 1) / (32L * tb));
  *n_partials = (int)blocks;
  ProfScope prof(JD_KERNEL_GMM_FWD, s);
  if (tri) {
    switch (tb) {
      case 16: return launch_fwd_tb<16, MODE, true>(a, blocks, s);
      case 8: return launch_fwd_tb<8, MODE, true>(a, blocks, s);
      default: return launch_fwd_tb<4, MODE, true>(a, blocks, s);
    }
  }
  switch (tb) {
    case 16: return launch_fwd_tb<16, MODE, false>(a, blocks, s);
    case 8: return launch_fwd_tb<8, MODE, false>(a, blocks, s);
    default: return launch_fwd_tb<4, MODE, false>(a, blocks, s);
  }
}

// Max mode through the fp16 screen (see gmm_screen_kernel): fills a.argmax_out (if any) and one fp64 partial sum per
// 1024 patches, exactly the numbers gmm_fwd_kernel<MODE_MAX> produces.
// fused: the exact kernel also writes the gradient row of every surviving record and gmm_best_kernel the winning row
// of every patch (g->grec, g->winner); after a fallback the components are in fallback_argmax instead.
// lse: logsumexp mode (always with the gradient; see gmm_lse_combine_kernel): the screen keeps the components within
// LSE_MARGIN of the lower bound, the exact kernel stores l per record, the combine kernel turns the records of a patch
// into its value and its gradient row (g->gpatch); the dense logsumexp kernels are enqueued behind the device flag.
static int screened_forward(jd_gmm* g, const GmmFwdArgs& a, hipStream_t s, int* n_partials, bool fused,
                            int32_t* fallback_argmax, double value_scale, float* value_out, int accumulate_value,
                            bool lse = false) {
  const long n = a.n_end - a.n_begin;
  // every wave its own 128 patches and all components, unless that leaves CUs without a block: then the four waves of
  // a block share 128 patches and split the components (see gmm_screen_kernel)
  bool ksplit = (n + SCREEN_T * 32 * 4 - 1) / (SCREEN_T * 32 * 4) < g->n_cu;
  if (opt_is_set(OPT_GMM_KSPLIT)) ksplit = opt_value(OPT_GMM_KSPLIT, 0) != 0;  // testing: force either decomposition
  // tuning: JD_GMM_SCREEN_NP=1 -- one tile pair (64 patches) per wave, two waves per SIMD (256 registers each)
  const bool np1 = !ksplit && opt_value(OPT_GMM_SCREEN_NP, 0) == 1 && g->K <= SCREEN_KC_MAX;
  const int T = np1 ? 2 : SCREEN_T;
  const unsigned blocks = (unsigned)(ksplit ? (n + T * 32 - 1) / (T * 32) : ((n + T * 32 - 1) / (T * 32) + 3) / 4);
  const size_t n_seg = (size_t)blocks * 4;
  const size_t slots = n_seg * SCREEN_CAP;                  // candidate record slots
  const size_t bucket_slots = slots + 32 * (size_t)g->K;    // padded bucket slots
  JD_REQUIRE(bucket_slots < (size_t)1 << 31, "jd_gmm_prior_fwd_bwd: too many patches for the screened path");
  int rc;
  if ((rc = grow(&g->best, &g->best_cap, (size_t)a.n_end))) return rc;
  if ((rc = grow(&g->lfinal, &g->lfinal_cap, (size_t)a.n_end))) return rc;
  if ((rc = grow(&g->rec, &g->rec_cap, 3 * slots))) return rc;
  if ((rc = grow(&g->rec_order, &g->rec_order_cap, bucket_slots))) return rc;
  if ((rc = grow(&g->rec_order_n, &g->rec_order_n_cap, bucket_slots))) return rc;
  if ((rc = grow(&g->seg_cnt, &g->seg_cnt_cap, n_seg))) return rc;
  if ((rc = grow(&g->partials, &g->partials_cap, (size_t)((n + 31) / 32 + 4)))) return rc;
  int32_t* rec_n = g->rec;
  int32_t* rec_k = g->rec + slots;
  float* rec_ub = reinterpret_cast<float*>(g->rec + 2 * slots);
  int* flag = g->screen_ctl;
  g->gen = g->gen % (1 << 30) + 1;
  // a pass whose shifts come from device memory may be REPLAYED from a captured graph with this very generation number:
  // a fallback flag left by an earlier replay must not be taken for this pass's (a node of the graph clears it)
  if (a.shift_dev) JD_HIP(hipMemsetAsync(flag, 0, sizeof(int), s));
  // rows of the record-gradient buffer: 1.0-1.3 records per patch survive on the seeded mixtures of the benchmark, up
  // to 1.9 on noise under an image-like mixture (condition numbers 1e5: wider bounds); beyond 4 per patch (+ bucket
  // padding; 1 KB per patch) the scan kernel raises the fallback flag and the dense kernel takes the pass
  if (fused && g->host_stats) {
    volatile int* hs = g->host_stats;
    const int seen = hs[0];
    if (seen != g->stats_seen_gen && seen > 0) {  // a pass has finished since the last look
      const long used = hs[2], patches = hs[3];
      if (patches > 0 && (hs[1] != 0 || used - 32L * g->K > (long)(0.6 * g->rows_per_patch * (double)patches)) && g->rows_per_patch < 32)
        g->rows_per_patch *= 2;
      g->stats_seen_gen = seen;
    }
  }
  const size_t grec_rows = fused ? (size_t)g->rows_per_patch * (size_t)n + 32 * (size_t)g->K : 0;
  if (fused) {
    if ((rc = grow(&g->grec, &g->grec_cap, grec_rows * D))) return rc;
    if ((rc = grow(&g->winner, &g->winner_cap, (size_t)a.n_end))) return rc;
  }
  const unsigned combine_blocks = (unsigned)((n + 15) / 16), value_blocks = (unsigned)((n + 1023) / 1024);
  if (lse) {
    if ((rc = grow(&g->lrec, &g->lrec_cap, bucket_slots))) return rc;
    if ((rc = grow(&g->pcount, &g->pcount_cap, (size_t)a.n_end))) return rc;
    if ((rc = grow(&g->dense_mark, &g->dense_mark_cap, (size_t)a.n_end))) return rc;
    if ((rc = grow(&g->vpatch, &g->vpatch_cap, (size_t)a.n_end))) return rc;
    if ((rc = grow(&g->ptab, &g->ptab_cap, (size_t)a.n_end * LSE_ROWS))) return rc;
    if ((rc = grow(&g->partials_lse, &g->partials_lse_cap, (size_t)value_blocks))) return rc;
    if ((rc = grow(&g->marked_lse, &g->marked_lse_cap, (size_t)value_blocks))) return rc;
    if ((rc = grow(&g->dense_list, &g->dense_list_cap, (size_t)n + 1))) return rc;
  }

  const size_t n_tiles = (size_t)blocks * (ksplit ? T : 4 * T);  // tiles the screen's waves touch
  if ((rc = grow(&g->xfrag, &g->xfrag_cap, n_tiles * 4 * 64))) return rc;
  if ((rc = grow(&g->xstat, &g->xstat_cap, 2 * n_tiles * 32))) return rc;
  if ((rc = grow(&g->xok, &g->xok_cap, n_tiles * 32))) return rc;

  // chunks of the record sort: one record segment per block unless there are too many (the kernels stride then)
  unsigned chunks = (unsigned)n_seg;
  unsigned max_blocks = std::max<unsigned>(2u * g->n_cu, (1u << 20) / (unsigned)g->K);
  if (opt_value(OPT_GMM_SORT_BLOCKS, 0) > 0) max_blocks = (unsigned)opt_value(OPT_GMM_SORT_BLOCKS, 0);
  if (chunks > max_blocks) chunks = max_blocks;
  if ((rc = grow(&g->blk_counts, &g->blk_counts_cap, (size_t)chunks * g->K))) return rc;
  const bool kc_lds = g->K <= SCREEN_KC_MAX && !opt_is_set(OPT_GMM_SCREEN_NO_LDS_CONSTS);  // (testing: the global-load path)

  ProfScope prof(JD_KERNEL_GMM_FWD, s);
  GmmStageArgs stg{};
  stg.flux = a.flux, stg.H = a.H, stg.W = a.W, stg.stride = a.stride, stg.nPx = a.nPx, stg.shift_y = a.shift_y, stg.shift_x = a.shift_x, stg.shift_dev = a.shift_dev;
  stg.n_begin = a.n_begin, stg.n_end = a.n_end, stg.n_tiles = (int)n_tiles;
  stg.pcount = lse ? g->pcount : nullptr, stg.dense_mark = lse ? g->dense_mark : nullptr;
  stg.dense_count = lse ? reinterpret_cast<int*>(g->dense_list) : nullptr;
  stg.xfrag = g->xfrag, stg.xn = g->xstat, stg.xs2 = g->xstat + n_tiles * 32, stg.ok = g->xok, stg.best = g->best;
  GmmScreenArgs sc{};
  sc.xfrag = g->xfrag, sc.xn = stg.xn, sc.xs2 = stg.xs2, sc.ok = g->xok;
  sc.flux = a.flux, sc.afrag16 = g->afrag16, sc.const_k = g->const_k, sc.efro_k = g->efro_k, sc.sk2_k = g->sk2_k, sc.mnorm_k = g->mnorm_k, sc.korder = g->korder;
  sc.K = a.K, sc.H = a.H, sc.W = a.W, sc.stride = a.stride, sc.nPx = a.nPx, sc.shift_y = a.shift_y, sc.shift_x = a.shift_x, sc.shift_dev = a.shift_dev;
  sc.n_begin = a.n_begin, sc.n_end = a.n_end;
  sc.lfinal = g->lfinal, sc.rec_n = rec_n, sc.rec_k = rec_k, sc.rec_ub = rec_ub;
  sc.seg_cnt = g->seg_cnt, sc.flag = flag, sc.gen = g->gen, sc.dense_mark = lse ? g->dense_mark : nullptr;
  {
    ProfScope stage(JD_KERNEL_GMM_STAGE, s);
    gmm_stage_kernel<<<(unsigned)((n_tiles + 3) / 4), 256, 0, s>>>(stg);
  }
  JD_LAUNCH_CHECK();
  {
    ProfScope stage(JD_KERNEL_GMM_SCREEN, s);
    if (lse && ksplit)  // (the caller has checked K <= SCREEN_KC_MAX: the constants table is in LDS)
      gmm_screen_kernel<2, true, true, true><<<blocks, 256, 0, s>>>(sc);
    else if (lse)
      gmm_screen_kernel<2, false, true, true><<<blocks, 256, 0, s>>>(sc);
    else if (np1)
      gmm_screen_kernel<1, false, true><<<blocks, 256, 0, s>>>(sc);
    else if (ksplit && kc_lds)
      gmm_screen_kernel<2, true, true><<<blocks, 256, 0, s>>>(sc);
    else if (ksplit)
      gmm_screen_kernel<2, true, false><<<blocks, 256, 0, s>>>(sc);
    else if (kc_lds && g->clock_stamps) {  // (the default instantiation with the clock stamps: jd_gmm_screen_clock)
      sc.clock_stamps = g->clock_stamps, sc.clock_cap = SCREEN_CLOCK_CAP;
      gmm_screen_kernel<2, false, true, false, true><<<blocks, 256, 0, s>>>(sc);
    } else if (kc_lds)
      gmm_screen_kernel<2, false, true><<<blocks, 256, 0, s>>>(sc);
    else
      gmm_screen_kernel<2, false, false><<<blocks, 256, 0, s>>>(sc);
  }
  JD_LAUNCH_CHECK();

  // counting sort of the surviving records by component (the record slot plays the role of the patch index)
  GmmBucketArgs bk{};
  bk.argmax = rec_k, bk.n_begin = 0, bk.n_end = (int)slots, bk.K = g->K;
  bk.counts = g->screen_ctl + 1, bk.offsets = g->screen_ctl + 1 + 2 * g->K;
  bk.order = g->rec_order, bk.order_n = g->rec_order_n, bk.gpatch = nullptr;
  bk.seg_cnt = g->seg_cnt, bk.seg_cap = SCREEN_CAP, bk.rec_n = rec_n, bk.rec_ub = rec_ub, bk.lfinal = g->lfinal;
  bk.chunk = SCREEN_CAP;  // one record segment per chunk
  bk.korder = g->K <= KORDER_MAX_K ? g->korder : nullptr;
  if (fused) bk.flag = flag, bk.gen = g->gen, bk.slot_cap = (int)std::min<size_t>(grec_rows, (size_t)INT32_MAX);
  if (lse) bk.margin = LSE_MARGIN, bk.pcount = g->pcount, bk.ptab = g->ptab, bk.ptab_rows = LSE_ROWS, bk.dense_mark = g->dense_mark;
  bk.blk_counts = g->blk_counts;
  const size_t hist_bytes = (size_t)g->K * sizeof(int);
  {
    ProfScope stage(JD_KERNEL_GMM_SORT, s);
    gmm_bucket_count_kernel<<<chunks, 256, hist_bytes, s>>>(bk);
    launch_binscan(bk, g->K, (int)chunks, s);
    gmm_bucket_scatter_kernel<<<chunks, 256, 3 * hist_bytes + sizeof(int), s>>>(bk);
  }
  JD_LAUNCH_CHECK();

  GmmExactArgs ex{};
  ex.flux = a.flux, ex.afrag = g->afrag, ex.mfrag = g->mfrag, ex.const_k = g->const_k;
  ex.order_n = g->rec_order_n, ex.counts = bk.counts, ex.offsets = bk.offsets, ex.flag = flag, ex.gen = g->gen;
  ex.gfrag = g->gfrag, ex.grec = fused ? g->grec : nullptr, ex.lrec = lse ? g->lrec : nullptr;
  ex.best = g->best, ex.K = g->K, ex.H = a.H, ex.W = a.W, ex.stride = a.stride, ex.nPx = a.nPx;
  ex.shift_y = a.shift_y, ex.shift_x = a.shift_x, ex.shift_dev = a.shift_dev;
#ifdef JD_EXACT_STAMPS
  static unsigned long long* stamps_dev = nullptr;
  static size_t stamps_cap = 0;
  const size_t max_groups = slots / 32 + g->K + 8;
  if (stamps_cap < max_groups) {
    if (stamps_dev) (void)hipFree(stamps_dev);
    JD_HIP(hipMalloc(&stamps_dev, max_groups * 8 * sizeof(unsigned long long)));
    stamps_cap = max_groups;
  }
  ex.stamps = stamps_dev;
#endif
  {
    ProfScope stage(JD_KERNEL_GMM_EXACT, s);
    gmm_exact_kernel<true><<<(unsigned)(g->n_cu * 3), 256, 0, s>>>(ex);
  }
  JD_LAUNCH_CHECK();
#ifdef JD_EXACT_STAMPS
  if (opt_is_set(OPT_GMM_SCREEN_DEBUG)) {  // phase histogram of this launch (synchronises)
    JD_HIP(hipStreamSynchronize(s));
    int total_slots = 0;
    JD_HIP(hipMemcpy(&total_slots, bk.offsets + g->K, sizeof(int), hipMemcpyDeviceToHost));
    const int n_groups = total_slots >> 5;
    std::vector<unsigned long long> st((size_t)n_groups * 8);
    JD_HIP(hipMemcpy(st.data(), stamps_dev, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    const char* names[6] = {"bucket + fragments", "indices + patch rows -> LDS", "LDS read-back + mean", "first product (80 MFMAs)",
                            "value epilogue + atomicMax", "second product + gradient rows"};
    double sum[6] = {0};
    std::vector<double> per[6];
    unsigned long long t_min = ~0ull, t_max = 0;
    for (int gi = 0; gi < n_groups; ++gi) {
      const unsigned long long* t = &st[(size_t)gi * 8];
      for (int ph = 0; ph < 6; ++ph) {
        const double dt = (double)(t[ph + 1] - t[ph]);
        sum[ph] += dt;
        per[ph].push_back(dt);
      }
      t_min = t[0] < t_min ? t[0] : t_min, t_max = t[6] > t_max ? t[6] : t_max;
    }
    fprintf(stderr, "[jd exact stamps] %d groups of 32 records, first stamp to last stamp %.0f shader cycles\n", n_groups,
            (double)(t_max - t_min));
    double total = 0;
    for (int ph = 0; ph < 6; ++ph) total += sum[ph];
    for (int ph = 0; ph < 6; ++ph) {
      std::sort(per[ph].begin(), per[ph].end());
      const size_t m = per[ph].size();
      fprintf(stderr, "[jd exact stamps] %-34s mean %8.0f cycles (%4.1f %%)  q10 %7.0f  q50 %7.0f  q90 %7.0f  q99 %7.0f\n", names[ph],
              sum[ph] / n_groups, 100.0 * sum[ph] / total, per[ph][m / 10], per[ph][m / 2], per[ph][m * 9 / 10], per[ph][m * 99 / 100]);
    }
    fprintf(stderr, "[jd exact stamps] per group %.0f cycles; groups per wave %.2f; waves %d\n", total / n_groups,
            (double)n_groups / (double)(g->n_cu * 12), g->n_cu * 12);
  }
#endif

  if (lse) {
    // the dense logsumexp kernel on the groups that hold a marked patch (after a fallback of the pass: on all of them),
    // the records of every other patch -> value and gradient row, the values summed in a fixed order
    GmmBwdLseArgs b{};
    b.flux = a.flux, b.afrag = g->afrag, b.mfrag = g->mfrag, b.gfrag = g->gfrag, b.const_k = g->const_k;
    b.partials = g->partials, b.gpatch = g->gpatch, b.K = g->K;
    b.H = a.H, b.W = a.W, b.stride = a.stride, b.nPx = a.nPx, b.shift_y = a.shift_y, b.shift_x = a.shift_x, b.shift_dev = a.shift_dev;
    b.n_begin = a.n_begin, b.n_end = a.n_end, b.run_flag = flag, b.run_gen = g->gen;
    b.mark = g->dense_mark, b.vpatch = g->vpatch;
    b.list = g->dense_list + 1, b.list_count = reinterpret_cast<const int*>(g->dense_list);
    gmm_lse_list_kernel<<<value_blocks, 256, 0, s>>>(g->dense_mark, a.n_begin, a.n_end, flag, g->gen, g->dense_list + 1,
                                                     reinterpret_cast<int*>(g->dense_list));
    JD_LAUNCH_CHECK();
    long bblocks = ((n + 31) / 32 + 2 * 4 - 1) / (2 * 4);
    if (bblocks > g->n_cu) bblocks = g->n_cu;
    gmm_bwd_lse_kernel<true, 2><<<(unsigned)bblocks, 256, 0, s>>>(b);
    JD_LAUNCH_CHECK();
    GmmLseCombineArgs cb{};
    cb.pcount = g->pcount, cb.ptab = g->ptab, cb.rows = LSE_ROWS, cb.lrec = g->lrec, cb.grec = g->grec, cb.gpatch = g->gpatch;
    cb.vpatch = g->vpatch, cb.mark = g->dense_mark, cb.n_begin = a.n_begin, cb.n_end = a.n_end, cb.flag = flag, cb.gen = g->gen;
    gmm_lse_combine_kernel<<<combine_blocks, 256, 0, s>>>(cb);
    JD_LAUNCH_CHECK();
    gmm_lse_value_kernel<<<value_blocks, 256, 0, s>>>(g->vpatch, g->dense_mark, a.n_begin, a.n_end, g->partials_lse, g->marked_lse);
    JD_LAUNCH_CHECK();
    gmm_lse_finalize_kernel<<<1, 256, 0, s>>>(g->partials_lse, g->marked_lse, (int)value_blocks, flag, g->gen, value_scale,
                                              value_out, accumulate_value, g->host_stats_dev, bk.offsets + g->K, (int)n);
    JD_LAUNCH_CHECK();
    *n_partials = 0;
    return JD_OK;
  }

  // fallback: the dense fp32 kernel, gated on the device flag (returns at once in the normal case)
  GmmFwdArgs dense = a;
  dense.run_flag = flag, dense.run_gen = g->gen, dense.best_out = g->best, dense.argmax_out = nullptr, dense.value_patch = nullptr;
  {
    const int tb = pick_block_tiles(n, g->n_cu);
    const unsigned dblocks = (unsigned)((n + 32L * tb - 1) / (32L * tb));
    switch (tb) {
      case 16: rc = launch_fwd_tb<16, MODE_MAX, true>(dense, dblocks, s); break;
      case 8: rc = launch_fwd_tb<8, MODE_MAX, true>(dense, dblocks, s); break;
      default: rc = launch_fwd_tb<4, MODE_MAX, true>(dense, dblocks, s); break;
    }
    if (rc) return rc;
  }

  GmmBestArgs be{};
  be.best = g->best, be.n_begin = a.n_begin, be.n_end = a.n_end, be.argmax_out = a.argmax_out, be.partials = g->partials;
  be.flag = flag, be.gen = g->gen, be.winner = fused ? g->winner : nullptr, be.argmax_fb = fused ? fallback_argmax : nullptr;
  be.rec_k = rec_k, be.rec_order = g->rec_order;
  be.ticket = g->screen_ctl + 3 * g->K + 2, be.scale = value_scale, be.value_out = value_out, be.accumulate = accumulate_value;
  be.host_stats = fused ? g->host_stats_dev : nullptr, be.slots_used = bk.offsets + g->K;
  if (fused) {
    GmmBwdFallbackArgs& b = be.fb;
    b.flux = a.flux, b.afrag = g->afrag, b.mfrag = g->mfrag, b.gfrag = g->gfrag, b.argmax = fallback_argmax, b.gpatch = g->gpatch;
    b.flag = flag, b.gen = g->gen, b.K = g->K;
    b.H = a.H, b.W = a.W, b.stride = a.stride, b.nPx = a.nPx, b.shift_y = a.shift_y, b.shift_x = a.shift_x, b.shift_dev = a.shift_dev;
    b.n_begin = a.n_begin, b.n_end = a.n_end;
  }
  const unsigned best_blocks = (unsigned)((n + BEST_CHUNK - 1) / BEST_CHUNK);
  gmm_best_kernel<<<best_blocks, 256, 0, s>>>(be);
  JD_LAUNCH_CHECK();
  *n_partials = (int)best_blocks;
  if (opt_is_set(OPT_GMM_SCREEN_DEBUG)) {  // tuning only: synchronises
    std::vector<int> ctl(3 * g->K + 2), seg(n_seg);
    JD_HIP(hipStreamSynchronize(s));
    JD_HIP(hipMemcpy(ctl.data(), g->screen_ctl, ctl.size() * sizeof(int), hipMemcpyDeviceToHost));
    JD_HIP(hipMemcpy(seg.data(), g->seg_cnt, seg.size() * sizeof(int), hipMemcpyDeviceToHost));
    long survivors = 0, records = 0;
    int seg_max = 0;
    for (int k = 0; k < g->K; ++k) survivors += ctl[1 + k];
    for (int v : seg) records += v, seg_max = v > seg_max ? v : seg_max;
    fprintf(stderr, "[jd gmm screen] patches %ld records %ld (%.2f per patch, fullest wave %d of %d) survivors %ld (%.2f per "
            "patch) fallback %d\n", n, records, (double)records / (double)n, seg_max, SCREEN_CAP, survivors,
            (double)survivors / (double)n, ctl[0] == g->gen ? 1 : 0);
  }
  return JD_OK;
}

static int gmm_prior_impl(jd_gmm* g, const float* flux, int H, int W, int stride, int shift_y,
                          int shift_x, int patch_row_begin, int patch_row_end, int marginalize,
                          float value_scale, float* value_out, int accumulate_value, float grad_coef,
                          float* grad_flux_accum, int32_t* argmax_out, float* band_out, void* stream,
                          const AdamArgs* step = nullptr, const int* shift_dev = nullptr, int phases = 3) {
  JD_REQUIRE(g && flux && value_out, "jd_gmm_prior_fwd_bwd: null argument");
  JD_REQUIRE(phases >= 1 && phases <= 3, "jd_gmm_prior_fwd_bwd: phases = %d not in [1, 3]", phases);
  JD_REQUIRE(H >= P && W >= P, "jd_gmm_prior_fwd_bwd: image (%d, %d) smaller than a patch", H, W);
  JD_REQUIRE(stride >= 1 && stride <= P, "jd_gmm_prior_fwd_bwd: stride = %d not in [1, 8]", stride);
  const int nPy = (H - P) / stride + 1, nPx = (W - P) / stride + 1;
  JD_REQUIRE((long)nPy * nPx < (1L << 31), "jd_gmm_prior_fwd_bwd: too many patches");
  if (patch_row_end < 0) patch_row_end = nPy;
  JD_REQUIRE(patch_row_begin >= 0 && patch_row_begin <= patch_row_end && patch_row_end <= nPy,
             "jd_gmm_prior_fwd_bwd: patch row range [%d, %d) outside [0, %d]", patch_row_begin, patch_row_end, nPy);
  hipStream_t s = as_stream(stream);
  shift_y = ((shift_y % H) + H) % H;  // roll by any integer = roll by its residue (the kernels' wrap() relies on it)
  shift_x = ((shift_x % W) + W) % W;
  const int n_begin = patch_row_begin * nPx, n_end = patch_row_end * nPx;
  if (n_begin == n_end) {  // empty shard: contributes nothing (an empty band has no rows)
    if (!accumulate_value) JD_HIP(hipMemsetAsync(value_out, 0, sizeof(float), s));
    return JD_OK;
  }
  if (band_out) grad_flux_accum = band_out;  // "a gradient is wanted"; the gather writes the band instead
  if (step) {
    JD_REQUIRE(!band_out && patch_row_begin == 0 && patch_row_end == nPy && stride >= 4 && opt_value(OPT_GMM_GATHER_TILED, 1) != 0,
               "jd_gmm_prior_fwd_bwd_step: needs the whole prior (no shard, no band) and stride >= 4");
    grad_flux_accum = step->grad_flux;  // "a gradient is wanted"; the gather reads it and applies the step instead
  }
  const long n = n_end - n_begin;
  int rc;
  // phases: bit 0 = everything up to the per-patch gradient rows (value, arg-max, rows: reads the flux only), bit 1 = the
  // gather (+ optimizer step) of those rows into the gradient image.  Called with 1 and later with 2 -- the same arguments --
  // the two halves may sit on different streams: the caller runs the first beside the likelihood launches of the step (it
  // does not touch the gradient image) and joins the streams in front of the second (jolideco_amd/core.py).
  bool screened = false, fused = false, lse_screened = false;
  if (phases & 1) {
    if ((rc = grow(&g->partials, &g->partials_cap, (size_t)((n + 31) / 32 + 4)))) return rc;
    // option JD_GMM_SCREEN = 0 forces the dense fp32 kernel (testing / tuning)
    screened = !marginalize && g->screen_ok && opt_value(OPT_GMM_SCREEN, 1) != 0 && !opt_is_set(OPT_GMM_DENSE);
    // screened arg-max with a gradient: the exact kernel also produces the gradient rows (no second sort, no separate
    // backward kernel); JD_GMM_FUSED_BWD=0 keeps the bucketed backward pass (testing / tuning)
    fused = screened && grad_flux_accum && g->triangular && opt_value(OPT_GMM_FUSED_BWD, 1) != 0;
    // logsumexp mode with a gradient: through the screen as well (option JD_GMM_LSE_SCREEN = 0: the dense kernels)
    lse_screened = marginalize && grad_flux_accum && g->screen_ok && g->triangular && g->K <= SCREEN_KC_MAX &&
                        opt_value(OPT_GMM_LSE_SCREEN, 1) != 0 && !opt_is_set(OPT_GMM_DENSE);
    if (lse_screened && g->host_stats && opt_value(OPT_GMM_LSE_SCREEN, 1) != 2) {  // (2: always, for tests and timing)
      volatile int* hs = g->host_stats;
      const int seen = hs[0];
      if (g->last_pass_lse && seen == g->gen && seen != g->lse_seen_gen) {  // the previous pass has landed and was screened
        g->lse_seen_gen = seen;
        if (hs[1] == 2 || (hs[1] == 1 && g->rows_per_patch >= 32)) g->lse_skip = 32;
      }
      if (g->lse_skip > 0) --g->lse_skip, lse_screened = false;
    }
    g->last_pass_lse = lse_screened;
    int32_t* arg = argmax_out;
    if (grad_flux_accum && (!arg || fused)) {  // fused: the internal buffer holds the components after a fallback
      if ((rc = grow(&g->argmax, &g->argmax_cap, (size_t)nPy * nPx))) return rc;
      if (!arg) arg = g->argmax;
    }
    GmmFwdArgs a{};
    a.flux = flux, a.afrag = g->afrag, a.mfrag = g->mfrag, a.const_k = g->const_k;
    a.K = g->K, a.H = H, a.W = W, a.stride = stride, a.nPx = nPx, a.shift_y = shift_y, a.shift_x = shift_x, a.shift_dev = shift_dev;
    a.n_begin = n_begin, a.n_end = n_end, a.argmax_out = fused ? argmax_out : arg, a.value_patch = nullptr, a.partials = g->partials;
    if (grad_flux_accum && (rc = grow(&g->gpatch, &g->gpatch_cap, (size_t)n * D))) return rc;  // (the fused fallback writes it)
    int n_waves = 0;
    if (lse_screened)
      rc = screened_forward(g, a, s, &n_waves, true, nullptr, (double)value_scale, value_out, accumulate_value, true);
    else if (marginalize && grad_flux_accum) {
      // value and gradient rows in one pass over the components (gmm_bwd_lse_kernel)
      GmmBwdLseArgs b{};
      b.flux = flux, b.afrag = g->afrag, b.mfrag = g->mfrag, b.gfrag = g->gfrag, b.const_k = g->const_k;
      b.partials = g->partials, b.gpatch = g->gpatch, b.K = g->K;
      b.H = H, b.W = W, b.stride = stride, b.nPx = nPx, b.shift_y = shift_y, b.shift_x = shift_x, b.shift_dev = shift_dev;
      b.n_begin = n_begin, b.n_end = n_end;
      const long groups = (n + 31) / 32;
      long blocks = (groups + 2 * 4 - 1) / (2 * 4);  // 2 groups per wave, 4 waves per block
      if (blocks > g->n_cu) blocks = g->n_cu;       // one block per CU (one wave per SIMD), grid-stride over the rest
      if ((rc = grow(&g->partials, &g->partials_cap, (size_t)blocks))) return rc;
      b.partials = g->partials;
      n_waves = (int)blocks;
      ProfScope prof(JD_KERNEL_GMM_BWD, s);
      if (g->triangular && !opt_is_set(OPT_GMM_DENSE))
        gmm_bwd_lse_kernel<true, 2><<<(unsigned)blocks, 256, 0, s>>>(b);
      else
        gmm_bwd_lse_kernel<false, 2><<<(unsigned)blocks, 256, 0, s>>>(b);
      JD_LAUNCH_CHECK();
    } else if (marginalize)
      rc = launch_fwd<MODE_LSE>(a, g->triangular, g->n_cu, s, &n_waves);
    else if (screened)
      rc = screened_forward(g, a, s, &n_waves, fused, fused ? g->argmax : nullptr, (double)value_scale, value_out, accumulate_value);
    else
      rc = launch_fwd<MODE_MAX>(a, g->triangular, g->n_cu, s, &n_waves);
    if (rc) return rc;
    // (screened path: the last block of gmm_best_kernel has written the value already)
    if (!screened && !lse_screened &&
        (rc = launch_finalize_sum(g->partials, n_waves, (double)value_scale, 0.0, value_out, accumulate_value, s)))
      return rc;
    if (!grad_flux_accum) return JD_OK;

    if ((rc = grow(&g->gpatch, &g->gpatch_cap, (size_t)n * D))) return rc;
    if (lse_screened) {
      // the combine kernel (or, after a fallback, the gated dense backward kernel) has written g->gpatch
    } else if (marginalize) {
      // gmm_bwd_lse_kernel has written the rows together with the value
    } else if (fused) {
      // the rows are in g->grec already (after a fallback: in g->gpatch, written by gmm_best_kernel's blocks)
    } else {
    const size_t slots_cap = (size_t)n + 32 * (size_t)g->K;
    if ((rc = grow(&g->order, &g->order_cap, slots_cap))) return rc;
    // ---- bucket the patches by arg-max component -------------------------------------------------
    GmmBucketArgs bk{};
    bk.argmax = arg, bk.n_begin = n_begin, bk.n_end = n_end, bk.K = g->K;
    bk.counts = g->bucket, bk.offsets = g->bucket + 2 * g->K;
    bk.order = g->order, bk.gpatch = g->gpatch;
    bk.chunk = BUCKET_CHUNK;
    unsigned chunks = (unsigned)((n + BUCKET_CHUNK - 1) / BUCKET_CHUNK);
    const unsigned max_blocks = std::max<unsigned>(2u * g->n_cu, (1u << 20) / (unsigned)g->K);
    if (chunks > max_blocks) chunks = max_blocks;
    if ((rc = grow(&g->blk_counts, &g->blk_counts_cap, (size_t)chunks * g->K))) return rc;
    bk.blk_counts = g->blk_counts;
    const size_t hist_bytes = (size_t)g->K * sizeof(int);
    {
      ProfScope prof(JD_KERNEL_GMM_BWD, s);
      gmm_bucket_count_kernel<<<chunks, 256, hist_bytes, s>>>(bk);
      launch_binscan(bk, g->K, (int)chunks, s);
      gmm_bucket_scatter_kernel<<<chunks, 256, 3 * hist_bytes + sizeof(int), s>>>(bk);
      GmmBwdArgs b{};
      b.flux = flux, b.afrag = g->afrag, b.mfrag = g->mfrag, b.gfrag = g->gfrag, b.argmax = arg, b.order = g->order;
      b.offsets = bk.offsets, b.counts = bk.counts, b.gpatch = g->gpatch, b.K = g->K;
      b.H = H, b.W = W, b.stride = stride, b.nPx = nPx, b.shift_y = shift_y, b.shift_x = shift_x, b.shift_dev = shift_dev;
      b.n_begin = n_begin, b.n_end = n_end;
      long bwd_blocks = ((long)(slots_cap / 32) + 3) / 4;
      const long cap = (long)g->n_cu * 3;  // 3 blocks of 4 waves per CU: one wave per SIMD x 3
      if (bwd_blocks > cap) bwd_blocks = cap;
      if (g->triangular && !opt_is_set(OPT_GMM_DENSE))
        gmm_bwd_max_kernel<true><<<(unsigned)bwd_blocks, 256, 0, s>>>(b);
      else
        gmm_bwd_max_kernel<false><<<(unsigned)bwd_blocks, 256, 0, s>>>(b);
    }
    JD_LAUNCH_CHECK();
    }

    g->pass = GmmPass{true, H, W, stride, shift_y, shift_x, patch_row_begin, patch_row_end, marginalize, fused, lse_screened, g->gen, shift_dev};
    if (!(phases & 2)) return JD_OK;
  } else {
    const GmmPass& ps = g->pass;
    JD_REQUIRE(ps.valid && ps.H == H && ps.W == W && ps.stride == stride && ps.shift_y == shift_y && ps.shift_x == shift_x &&
                   ps.row_begin == patch_row_begin && ps.row_end == patch_row_end && ps.marginalize == marginalize &&
                   ps.gen == g->gen && ps.shift_dev == shift_dev && grad_flux_accum,
               "jd_gmm_prior_fwd_bwd: phase 2 (gather) without the matching phase 1 of the same pass");
    fused = ps.fused, lse_screened = ps.lse_screened;
  }
  g->pass.valid = false;
  GmmGatherArgs ga{};
  ga.gpatch = g->gpatch, ga.grad = grad_flux_accum, ga.H = H, ga.W = W, ga.stride = stride, ga.nPx = nPx, ga.nPy = nPy;
  ga.shift_y = shift_y, ga.shift_x = shift_x, ga.shift_dev = shift_dev, ga.row_begin = patch_row_begin, ga.row_end = patch_row_end;
  ga.y_begin = patch_row_begin * stride;
  ga.y_end = (patch_row_end - 1) * stride + P;
  ga.coef = grad_coef;
  ga.band = band_out;
  if (fused) ga.winner = g->winner, ga.grec = g->grec, ga.flag = g->screen_ctl, ga.gen = g->gen;
  {
    auto aligned = [](const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15) == 0; };
    ga.vec = W % 4 == 0 && aligned(grad_flux_accum) ? 1 : 0;
    if (step) {
      ga.do_step = 1, ga.step = *step, ga.y_begin = 0, ga.y_end = H;  // every pixel of the image takes the step
      ga.preload = opt_value(OPT_GMM_GATHER_PRELOAD, 1) != 0;
      ga.vec = ga.vec && aligned(step->theta) && aligned(step->flux_in) && aligned(step->flux_out) && aligned(step->m) &&
               aligned(step->v) && aligned(step->mask);
    }
    ProfScope prof(JD_KERNEL_GMM_GATHER, s);
    // option JD_GMM_GATHER_TILED = 0: the per-pixel kernel (testing)
    if (stride >= 4 && opt_value(OPT_GMM_GATHER_TILED, 1) != 0) {
      // (x: the first tile starts up to 3 pixels left of the image so that the pixel groups are aligned un-rolled)
      dim3 grid((W + 3 + GATHER_T - 1) / GATHER_T, (ga.y_end - ga.y_begin + GATHER_T - 1) / GATHER_T);
      gmm_gather_tile_kernel<<<grid, 256, 0, s>>>(ga);
    } else {
      dim3 grid((W + 255) / 256, ga.y_end - ga.y_begin);
      gmm_gather_kernel<<<grid, 256, 0, s>>>(ga);
    }
  }
  JD_LAUNCH_CHECK();
  return JD_OK;
}

extern "C" int jd_gmm_prior_fwd_bwd(jd_gmm* g, const float* flux, int H, int W, int stride, int shift_y,
                                    int shift_x, int patch_row_begin, int patch_row_end, int marginalize,
                                    float value_scale, float* value_out, int accumulate_value, float grad_coef,
                                    float* grad_flux_accum, int32_t* argmax_out, const int* shift_dev, int phases,
                                    void* stream) {
  return gmm_prior_impl(g, flux, H, W, stride, shift_y, shift_x, patch_row_begin, patch_row_end, marginalize, value_scale,
                        value_out, accumulate_value, grad_coef, grad_flux_accum, argmax_out, nullptr, stream, nullptr, shift_dev,
                        phases);
}

// Diagnostics of the screened arg-max path (no synchronisation: whatever pass has landed in the host-mapped block):
// out = {generation of that pass, it fell back to the dense kernel (0 / 1), bucket slots it used, patches it covered,
// gradient rows per patch the record buffer currently has room for}.
extern "C" int jd_gmm_screen_stats(const jd_gmm* g, int* out) {
  JD_REQUIRE(g && out, "jd_gmm_screen_stats: null argument");
  for (int i = 0; i < 4; ++i) out[i] = g->host_stats ? reinterpret_cast<volatile int*>(g->host_stats)[i] : 0;
  out[4] = g->rows_per_patch;
  return JD_OK;
}

// The shader clock INSIDE the screen kernel (round-4 verdict: is the kernel short of its roof, or is the roof lower than the
// nominal clock says?).  First call: allocates the stamp buffer and switches the default screen launch of this handle to
// its stamped instantiation; every later call synchronises the device, averages 100 MHz x (shader ticks / reference
// ticks) over the blocks that have left stamps since the last call, and clears them.
extern "C" int jd_gmm_screen_clock(jd_gmm* g, double* mhz_out, int* samples_out) {
  JD_REQUIRE(g && mhz_out && samples_out, "jd_gmm_screen_clock: null argument");
  *mhz_out = 0.0, *samples_out = 0;
  const size_t bytes = (size_t)2 * SCREEN_CLOCK_CAP * sizeof(unsigned long long);
  if (!g->clock_stamps) {
    JD_HIP(hipMalloc(&g->clock_stamps, bytes));
    JD_HIP(hipMemset(g->clock_stamps, 0, bytes));
    return JD_OK;
  }
  std::vector<unsigned long long> host((size_t)2 * SCREEN_CLOCK_CAP);
  JD_HIP(hipDeviceSynchronize());
  JD_HIP(hipMemcpy(host.data(), g->clock_stamps, bytes, hipMemcpyDeviceToHost));
  JD_HIP(hipMemset(g->clock_stamps, 0, bytes));
  double sum = 0.0;
  int n = 0;
  for (int b = 0; b < SCREEN_CLOCK_CAP; ++b)
    if (host[2 * b + 1] > 0) sum += 100.0 * (double)host[2 * b] / (double)host[2 * b + 1], ++n;
  *samples_out = n;
  if (n) *mhz_out = sum / n;
  return JD_OK;
}

extern "C" int jd_gmm_prior_fwd_bwd_step(jd_gmm* g, const float* flux, int H, int W, int stride, int shift_y, int shift_x,
                                         int marginalize, float value_scale, float* value_out, int accumulate_value,
                                         float grad_coef, const jd_step* step, const int* shift_dev, int phases,
                                         void* stream) {
  JD_REQUIRE(step && step->theta && step->flux_in && step->flux_out && step->grad_flux, "jd_gmm_prior_fwd_bwd_step: null argument");
  JD_REQUIRE(step->sgd || (step->exp_avg && step->exp_avg_sq), "jd_gmm_prior_fwd_bwd_step: Adam needs its moment images");
  AdamArgs a{};
  a.theta = step->theta, a.flux_in = step->flux_in, a.flux_out = step->flux_out, a.grad_flux = const_cast<float*>(step->grad_flux);
  a.m = step->exp_avg, a.v = step->exp_avg_sq, a.mask = step->mask, a.n = (size_t)H * W;
  a.step_size = step->step_size, a.beta1 = step->beta1, a.beta2 = step->beta2, a.one_minus_beta1 = step->one_minus_beta1;
  a.one_minus_beta2 = step->one_minus_beta2, a.bias2_sqrt = step->bias2_sqrt, a.eps = step->eps, a.lr = step->lr;
  a.zero_grad = 0, a.sgd = step->sgd ? 1 : 0, a.linear = step->use_log_flux ? 0 : 1, a.bias_dev = step->bias_dev;
  return gmm_prior_impl(g, flux, H, W, stride, shift_y, shift_x, 0, -1, marginalize, value_scale, value_out, accumulate_value,
                        grad_coef, nullptr, nullptr, nullptr, stream, &a, shift_dev, phases);
}

extern "C" int jd_gmm_prior_band_fwd_bwd(jd_gmm* g, const float* flux, int H, int W, int stride, int shift_y,
                                         int shift_x, int patch_row_begin, int patch_row_end, int marginalize,
                                         float value_scale, float* value_out, int accumulate_value, float grad_coef,
                                         float* band_out, void* stream) {
  JD_REQUIRE(band_out, "jd_gmm_prior_band_fwd_bwd: null band");
  return gmm_prior_impl(g, flux, H, W, stride, shift_y, shift_x, patch_row_begin, patch_row_end, marginalize, value_scale,
                        value_out, accumulate_value, grad_coef, nullptr, nullptr, band_out, stream);
}

extern "C" int jd_add_rolled_bands(float* grad, int H, int W, int shift_y, int shift_x, const float* bands,
                                   size_t chunk_floats, int n_bands, const int* y_begin, const int* y_end, void* stream) {
  JD_REQUIRE(grad && bands && y_begin && y_end, "jd_add_rolled_bands: null argument");
  JD_REQUIRE(n_bands >= 1 && n_bands <= BANDS_MAX, "jd_add_rolled_bands: %d bands not in [1, %d]", n_bands, BANDS_MAX);
  AddBandsArgs a{};
  a.grad = grad, a.bands = bands, a.chunk = chunk_floats, a.H = H, a.W = W, a.n_bands = n_bands;
  a.shift_y = ((shift_y % H) + H) % H, a.shift_x = ((shift_x % W) + W) % W;
  a.y_lo = H, a.y_hi = 0;
  for (int b = 0; b < n_bands; ++b) {
    JD_REQUIRE(y_begin[b] >= 0 && y_begin[b] <= y_end[b] && y_end[b] <= H && (size_t)(y_end[b] - y_begin[b]) * W <= chunk_floats,
               "jd_add_rolled_bands: band %d rows [%d, %d) do not fit", b, y_begin[b], y_end[b]);
    a.y_begin[b] = y_begin[b], a.y_end[b] = y_end[b];
    if (y_begin[b] < y_end[b]) a.y_lo = std::min(a.y_lo, y_begin[b]), a.y_hi = std::max(a.y_hi, y_end[b]);
  }
  if (a.y_lo >= a.y_hi) return JD_OK;
  dim3 grid((W + 1023) / 1024, a.y_hi - a.y_lo);
  if (n_bands <= 8) add_rolled_bands_kernel<8><<<grid, 256, 0, as_stream(stream)>>>(narrow_bands<8>(a));
  else if (n_bands <= 16) add_rolled_bands_kernel<16><<<grid, 256, 0, as_stream(stream)>>>(narrow_bands<16>(a));
  else add_rolled_bands_kernel<BANDS_MAX><<<grid, 256, 0, as_stream(stream)>>>(a);
  JD_LAUNCH_CHECK();
  return JD_OK;
}

extern "C" int jd_add_rolled_bands_step(int H, int W, int shift_y, int shift_x, const float* bands, size_t chunk_floats,
                                        int n_bands, const int* y_begin, const int* y_end, const jd_step* step, void* stream) {
  JD_REQUIRE(bands && y_begin && y_end && step, "jd_add_rolled_bands_step: null argument");
  JD_REQUIRE(step->theta && step->flux_in && step->flux_out && step->grad_flux, "jd_add_rolled_bands_step: null image");
  JD_REQUIRE(step->sgd || (step->exp_avg && step->exp_avg_sq), "jd_add_rolled_bands_step: Adam needs its moment images");
  JD_REQUIRE(n_bands >= 1 && n_bands <= BANDS_MAX, "jd_add_rolled_bands_step: %d bands not in [1, %d]", n_bands, BANDS_MAX);
  auto aligned = [](const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15) == 0; };
  JD_REQUIRE(W % 4 == 0 && aligned(step->theta) && aligned(step->flux_in) && aligned(step->flux_out) && aligned(step->grad_flux) &&
                 aligned(step->exp_avg) && aligned(step->exp_avg_sq) && aligned(step->mask),
             "jd_add_rolled_bands_step: needs W %% 4 == 0 and 16-byte aligned images (use jd_add_rolled_bands + jd_adam_step)");
  AddBandsArgs a{};
  a.grad = nullptr, a.bands = bands, a.chunk = chunk_floats, a.H = H, a.W = W, a.n_bands = n_bands;
  a.shift_y = ((shift_y % H) + H) % H, a.shift_x = ((shift_x % W) + W) % W;
  for (int b = 0; b < n_bands; ++b) {
    JD_REQUIRE(y_begin[b] >= 0 && y_begin[b] <= y_end[b] && y_end[b] <= H && (size_t)(y_end[b] - y_begin[b]) * W <= chunk_floats,
               "jd_add_rolled_bands_step: band %d rows [%d, %d) do not fit", b, y_begin[b], y_end[b]);
    a.y_begin[b] = y_begin[b], a.y_end[b] = y_end[b];
  }
  AdamArgs st{};
  st.theta = step->theta, st.flux_in = step->flux_in, st.flux_out = step->flux_out, st.grad_flux = const_cast<float*>(step->grad_flux);
  st.m = step->exp_avg, st.v = step->exp_avg_sq, st.mask = step->mask, st.n = (size_t)H * W;
  st.step_size = step->step_size, st.beta1 = step->beta1, st.beta2 = step->beta2, st.one_minus_beta1 = step->one_minus_beta1;
  st.one_minus_beta2 = step->one_minus_beta2, st.bias2_sqrt = step->bias2_sqrt, st.eps = step->eps, st.lr = step->lr;
  st.zero_grad = 0, st.sgd = step->sgd ? 1 : 0, st.linear = step->use_log_flux ? 0 : 1, st.bias_dev = step->bias_dev;
  dim3 grid((W + 1023) / 1024, H);
  ProfScope prof(JD_KERNEL_ADAM, as_stream(stream));
  if (n_bands <= 8) add_rolled_bands_step_kernel<8><<<grid, 256, 0, as_stream(stream)>>>(narrow_bands<8>(a), st);
  else if (n_bands <= 16) add_rolled_bands_step_kernel<16><<<grid, 256, 0, as_stream(stream)>>>(narrow_bands<16>(a), st);
  else add_rolled_bands_step_kernel<BANDS_MAX><<<grid, 256, 0, as_stream(stream)>>>(a, st);
  JD_LAUNCH_CHECK();
  return JD_OK;
}

extern "C" int jd_gmm_estimate_log_prob(jd_gmm* g, const float* x, int n, float* out, void* stream) {
  JD_REQUIRE(g && x && out && n > 0, "jd_gmm_estimate_log_prob: null argument or n <= 0");
  hipStream_t s = as_stream(stream);
  int rc;
  if ((rc = grow(&g->partials, &g->partials_cap, (size_t)((n + 31) / 32 + 4)))) return rc;
  GmmFwdArgs a{};
  a.flux = x, a.afrag = g->afrag, a.mfrag = g->mfrag, a.const_k = g->const_k;
  a.K = g->K, a.n_begin = 0, a.n_end = n, a.value_patch = out, a.partials = g->partials;
  int n_waves = 0;
  return launch_fwd<MODE_DENSE>(a, g->triangular, g->n_cu, s, &n_waves);
}
