// Direct "same" convolution / correlation with a small PSF on the fp32 matrix cores (gfx950).
//
//   out[y][x] = sum_{dy,dx} psf[dy][dx] * u[y + oy - dy][x + ox - dx],   u = image [* scale], 0 outside
//
// is, for one PSF row dy and a strip of 16 output columns, a banded (Toeplitz) 16 x KC matrix applied
// to KC = 16 + kw - 1 input columns of image row y + oy - dy.  With M = output column in the strip,
// N = output row and K = input column this is D[m][n] += A_dy[m][c] * B_dy[c][n] on
// v_mfma_f32_16x16x4_f32 (exact fp32: an fmaf chain in (dy, c) order, so the result is at least as
// accurate as the reference's FFT convolution, jolideco/utils/torch.py:347-370).  A_dy (the PSF in
// MFMA fragment order) is the same for every tile and is prepared once per (dataset, component) by
// jd_conv_psf_spectrum; B comes from an LDS-staged window of the image (halo included, zero padded),
// so the padded grid of the FFT path, K1 (pad * exposure) and K5 (exposure * crop of the
// correlation) are all folded into this kernel's load / store.
//
// Work split: block = 4 waves = 64 x 64 outputs; wave w owns the 16-column strip w and 4
// accumulators (4 x 16 rows).  Per (dy, k-step) one A fragment is reused by the 4 accumulators and
// each MFMA needs one conflict-free ds_read_b32 (row pitch == 2 mod 4).  Cost: kh * KC/4 * 4 MFMAs of
// 32 cycles per 1024 outputs, i.e. 17 cycles/output at 17x17 (~30 us for 2048^2 on 1024 SIMDs)
// against ~150 us for two rocFFT transforms + the k-space multiply; the FFT path stays for large PSFs.
//
// SPLIT (the default where it fits): the same Toeplitz product on the fp16 matrix cores with both operands split in
// two, x = x_hi + 2^-11 x_lo (x_hi = fp16(x), x_lo = fp16(2^11 (x - x_hi)): 22 significant bits; the factor 2^11 keeps
// x_lo a NORMAL fp16 number wherever x_hi is one, so a pixel 10^8 times fainter than the brightest of its tile still
// has all its bits), and three products per step in two accumulators,
//   D_main += A_hi B_hi,   D_cross += A_lo B_hi + A_hi B_lo,   D = D_main + 2^-11 D_cross
//   (v_mfma_f32_16x16x32_f16: K = 32 input columns per instruction)
// with fp32 accumulation: A_lo B_lo (2^-22 of a term) is dropped, so a term is off by <= 3 x 2^-22 = 7e-7 of itself
// and a sum of same-signed terms by no more (typically 1e-7; the reference's FFT convolution is no closer to the exact
// sum).  The window is scaled per tile by the power of two that puts its largest |value| into [2^13, 2^14) (exact;
// fp16 would otherwise overflow / lose the faint pixels), the PSF likewise once per table; both scales are undone
// exactly in the epilogue.  Cost: kh * 12 MFMAs of 16 cycles per 1024 outputs instead of kh * 32 of 32 cycles --
// 5.3 x less matrix time, so the kernel is bound by its LDS reads and the image traffic, not by the matrix cores.
#include <cstdint>
#include <cstdlib>

#include "jd_common.h"
#include "kernels.h"

namespace jd {

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int TILE = 64;  // outputs per block edge

struct DirectConvArgs {
  const float* in;         // (H, W)
  const float* in_scale;   // nullable, multiplied onto `in` while staging (forward: exposure)
  const float* afrag;      // kh * (KC/4) * 64 floats: Toeplitz fragments of the (possibly flipped) PSF
                           // SPLIT: kh * (KC/32) * 2 planes * 64 lanes * 8 fp16 of psf * s_p, then 1 / s_p (a float)
  float* out;              // (H, W)
  const float* out_scale;  // nullable, multiplied onto the result (adjoint: exposure)
  int H, W, kh;
  int oy;     // output row y reads input rows y + oy - dy
  int ox_in;  // first input column of the strip starting at x0 is x0 + ox_in  (= ox - (kw-1))
  float coef;
  int accumulate;
  // POISSON kernels (forward model of ONE component, no up-sampling): the epilogue is the Poisson pass -- `out`
  // receives g = d loss / d conv instead of the convolution, one fp64 loss partial per tile
  const float* background;  // (H, W)
  const float* counts;      // (H, W)
  float* npred_out;         // nullable (H, W)
  double* partials;         // n_tiles
  float eps, inv_n;
  int write_grad;
};

// Persistent, software pipelined: a block walks over its tiles; while the MFMAs of tile i run out of
// LDS (image window AND Toeplitz fragments, so the matrix stream never waits on vmcnt), the loads of
// tile i+1's window are already in flight into registers and are written to LDS after the epilogue.
// Only the first window load and the last epilogue of a block are not overlapped with MFMA work.
// POISSON: clip, + background, Poisson NLL and its gradient applied to the accumulators (models/npred.py:191,254-261;
// loss.py:35-37) -- the convolution never goes to memory and the stand-alone Poisson launch disappears.
template <int KC, bool VEC, bool POISSON = false, bool SPLIT = false>
__global__ __launch_bounds__(256) void direct_conv_kernel(DirectConvArgs a) {
  constexpr int STEPS = KC / 4;
  constexpr int KS = KC / 32;      // SPLIT: 32-column steps of the fp16 MFMA (KC = 32 or 64)
  constexpr int COLS = 48 + KC;    // window columns: 64 outputs + KC - 16 halo
  constexpr int PITCH = COLS + 2;  // == 2 (mod 4): the 16 rows x 2 columns of a half-wave hit 32 banks
  constexpr int PH = COLS + 8;     // SPLIT: fp16 per row of one plane (16-byte aligned rows; 44 / 60 words: rows 0-7 of a
                                   // ds_read_b128 group start on 8 different multiples of 4 banks)
  constexpr int NPF = (96 * COLS + 255) / 256;  // window elements per thread (kh <= 33: at most 96 rows)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int rows = TILE - 1 + a.kh;
  float* win = lds;                        // rows * PITCH
  float* afl = lds + ((rows * PITCH + 3) & ~3);  // kh * STEPS * 64 Toeplitz fragments
  // SPLIT: two fp16 planes of the window (hi, lo), then the fragment table (uint4 per lane, plane and step)
  _Float16* winH = reinterpret_cast<_Float16*>(lds);
  _Float16* winL = winH + rows * PH;
  uint4* afl16 = reinterpret_cast<uint4*>(lds + ((rows * PH + 3) & ~3));  // 2 planes x rows x PH halfs = rows * PH floats

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int total = rows * COLS;
  const int tiles_x = (a.W + TILE - 1) / TILE, tiles_y = (a.H + TILE - 1) / TILE;
  const int n_tiles = tiles_x * tiles_y;

  // Toeplitz fragments -> LDS (once per block; the table is a multiple of 256 floats)
  float inv_sp = 1.f;
  if constexpr (SPLIT) {
    const uint4* src = reinterpret_cast<const uint4*>(a.afrag);
    for (int i = threadIdx.x; i < a.kh * KS * 128; i += 256) afl16[i] = src[i];
    inv_sp = a.afrag[(size_t)a.kh * KS * 512];
  } else {
    const float4* src = reinterpret_cast<const float4*>(a.afrag);
    float4* dst = reinterpret_cast<float4*>(afl);
    for (int i = threadIdx.x; i < a.kh * STEPS * 16; i += 256) dst[i] = src[i];
  }

  // Prefetch registers of the NEXT tile's window.  VEC: the window columns start on a 16-byte
  // boundary of the image rows (W % 4 == 0, ox_in % 4 == 0), so a thread moves float4 chunks that are
  // entirely inside or entirely outside the image.  The (row, column) of a thread's chunks is fixed for
  // the whole kernel; tiles whose window lies inside the image take a predicate-free path.
  constexpr int C4 = COLS / 4;
  constexpr int NPF4 = (96 * C4 + 255) / 256;
  float4 pv4[VEC ? NPF4 : 1], ps4[VEC ? NPF4 : 1];
  float pv[VEC ? 1 : NPF], ps[VEC ? 1 : NPF];
  const int total4 = rows * C4;
  int wr[VEC ? NPF4 : 1], wc[VEC ? NPF4 : 1];
  if constexpr (VEC) {
#pragma unroll
    for (int u = 0; u < NPF4; ++u) {
      const int i = threadIdx.x + 256 * u;
      wr[u] = i < total4 ? i / C4 : -1;  // -1: this thread has no chunk u
      wc[u] = i < total4 ? (i - (i / C4) * C4) * 4 : 0;
    }
  }
  auto issue_window_loads = [&](int tile) {
    const int x0 = (tile % tiles_x) * TILE, y0 = (tile / tiles_x) * TILE;
    const int yin0 = y0 + a.oy - (a.kh - 1), xin0 = x0 + a.ox_in;
    if constexpr (VEC) {
      const bool interior = yin0 >= 0 && yin0 + rows <= a.H && xin0 >= 0 && xin0 + COLS <= a.W;
      if (interior) {
        const float* base = a.in + (size_t)yin0 * a.W + xin0;
        const float* sbase = a.in_scale ? a.in_scale + (size_t)yin0 * a.W + xin0 : nullptr;
#pragma unroll
        for (int u = 0; u < NPF4; ++u) {
          const int off = (wr[u] < 0 ? 0 : wr[u]) * a.W + wc[u];  // threads without a chunk re-read chunk 0
          pv4[u] = *reinterpret_cast<const float4*>(base + off);
          ps4[u] = sbase ? *reinterpret_cast<const float4*>(sbase + off) : make_float4(1.f, 1.f, 1.f, 1.f);
        }
      } else {
#pragma unroll
        for (int u = 0; u < NPF4; ++u) {
          const int y = yin0 + wr[u], x = xin0 + wc[u];
          const bool inside = wr[u] >= 0 && y >= 0 && y < a.H && x >= 0 && x < a.W;
          const size_t off = inside ? (size_t)y * a.W + x : 0;
          pv4[u] = inside ? *reinterpret_cast<const float4*>(a.in + off) : make_float4(0.f, 0.f, 0.f, 0.f);
          ps4[u] = (inside && a.in_scale) ? *reinterpret_cast<const float4*>(a.in_scale + off)
                                          : make_float4(1.f, 1.f, 1.f, 1.f);
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < NPF; ++u) {
        const int i = threadIdx.x + 256 * u;
        const int r = i / COLS, c = i - r * COLS;
        const int y = yin0 + r, x = xin0 + c;
        const bool inside = (i < total) && y >= 0 && y < a.H && x >= 0 && x < a.W;
        const size_t off = inside ? (size_t)y * a.W + x : 0;
        pv[u] = inside ? a.in[off] : 0.f;
        ps[u] = (inside && a.in_scale) ? a.in_scale[off] : 1.f;
      }
    }
  };
  // SPLIT: window * in_scale, scaled by the power of two `s` that puts the tile's largest |value| into [2^13, 2^14),
  // as two fp16 planes; returns 1 / s.  (One more barrier per tile for the block-wide maximum.)
  __shared__ float red_max[4];
  auto store_window_split = [&]() -> float {
    float m = 0.f;
    if constexpr (VEC) {
#pragma unroll
      for (int u = 0; u < NPF4; ++u) {
        pv4[u].x *= ps4[u].x, pv4[u].y *= ps4[u].y, pv4[u].z *= ps4[u].z, pv4[u].w *= ps4[u].w;
        if (wr[u] >= 0) m = fmaxf(fmaxf(m, fmaxf(fabsf(pv4[u].x), fabsf(pv4[u].y))), fmaxf(fabsf(pv4[u].z), fabsf(pv4[u].w)));
      }
    } else {
#pragma unroll
      for (int u = 0; u < NPF; ++u) {
        pv[u] *= ps[u];
        if (threadIdx.x + 256 * u < total) m = fmaxf(m, fabsf(pv[u]));
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (lane == 0) red_max[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red_max[0], red_max[1]), fmaxf(red_max[2], red_max[3]));
    int ex = 14;  // zero, inf or NaN somewhere: no scaling (non-finite values propagate through the fp16 planes)
    if (m > 0.f && m < 3.0e38f) (void)frexpf(m, &ex);
    ex = ex < -100 ? -100 : ex;
    const float s = ldexpf(1.f, 14 - ex);
    auto split = [&](float x, _Float16& hi, _Float16& lo) {
      const float xs = x * s;
      hi = (_Float16)xs;
      lo = (_Float16)((xs - (float)hi) * 2048.f);  // |xs - hi| <= 2^-11 |hi|: the scaled remainder is as large as hi
    };
    if constexpr (VEC) {
#pragma unroll
      for (int u = 0; u < NPF4; ++u) {
        if (wr[u] >= 0) {  // 4 fp16 = 8 bytes per plane (PH and wc are multiples of 4)
          _Float16 h[4], l[4];
          split(pv4[u].x, h[0], l[0]), split(pv4[u].y, h[1], l[1]), split(pv4[u].z, h[2], l[2]), split(pv4[u].w, h[3], l[3]);
          *reinterpret_cast<f16x4*>(winH + wr[u] * PH + wc[u]) = f16x4{h[0], h[1], h[2], h[3]};
          *reinterpret_cast<f16x4*>(winL + wr[u] * PH + wc[u]) = f16x4{l[0], l[1], l[2], l[3]};
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < NPF; ++u) {
        const int i = threadIdx.x + 256 * u;
        const int r = i / COLS, c = i - r * COLS;
        if (i < total) split(pv[u], winH[r * PH + c], winL[r * PH + c]);
      }
    }
    return ldexpf(1.f, ex - 14);
  };
  auto store_window = [&]() {
    if constexpr (VEC) {
#pragma unroll
      for (int u = 0; u < NPF4; ++u) {
        if (wr[u] >= 0) {  // PITCH is even: 8-byte aligned pairs
          float2* dst = reinterpret_cast<float2*>(win + wr[u] * PITCH + wc[u]);
          dst[0] = make_float2(pv4[u].x * ps4[u].x, pv4[u].y * ps4[u].y);
          dst[1] = make_float2(pv4[u].z * ps4[u].z, pv4[u].w * ps4[u].w);
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < NPF; ++u) {
        const int i = threadIdx.x + 256 * u;
        const int r = i / COLS, c = i - r * COLS;
        if (i < total) win[r * PITCH + c] = pv[u] * ps[u];
      }
    }
  };

  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share an
  // L2), so work item w goes to tile (w % 8) * chunk + w / 8: every XCD walks a contiguous band of tile
  // rows and the window halos of neighbouring tiles hit in ITS L2.  Placement only changes speed.
  const int chunk = (n_tiles + 7) / 8;
  const int n_work = 8 * chunk;
  auto tile_of = [&](int w) { return (w & 7) * chunk + (w >> 3); };  // may be >= n_tiles: no work
  auto next_work = [&](int w) {
    while (w < n_work && tile_of(w) >= n_tiles) w += gridDim.x;
    return w;
  };
  int work = next_work(blockIdx.x);
  int tile = work < n_work ? tile_of(work) : n_tiles;
  if (tile < n_tiles) issue_window_loads(tile);
  const int n = lane & 15, kk = lane >> 4;
  while (tile < n_tiles) {
    float inv = 1.f;  // SPLIT: undoes the window's and the PSF's power-of-two scales
    if constexpr (SPLIT)
      inv = store_window_split() * inv_sp;
    else
      store_window();
    __syncthreads();
    work = next_work(work + gridDim.x);
    const int next = work < n_work ? tile_of(work) : n_tiles;
    if (next < n_tiles) issue_window_loads(next);  // in flight during the MFMA phase below

    // this lane's outputs: out[y0 + 16 b + n][x0 + 16 wave + 4 kk + 0..3]
    const int x0 = (tile % tiles_x) * TILE, y0 = (tile / tiles_x) * TILE;
    const int x = x0 + wave * 16 + 4 * kk;
    const bool vec = (a.W % 4 == 0) && (x + 3 < a.W);
    // SPLIT: the matrix phase is short, so the operands of the epilogue -- (background, counts) or (out_scale, out) -- are
    // requested before it instead of after it
    float4 e0[SPLIT ? 4 : 1], e1[SPLIT ? 4 : 1];
    if constexpr (SPLIT) {
      if (vec) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int y = y0 + 16 * b + n;
          const size_t off = (size_t)(y < a.H ? y : y0) * a.W + x;  // row y0 always exists
          if constexpr (POISSON) {
            e0[b] = *reinterpret_cast<const float4*>(a.background + off);
            e1[b] = *reinterpret_cast<const float4*>(a.counts + off);
          } else {
            e0[b] = a.out_scale ? *reinterpret_cast<const float4*>(a.out_scale + off) : make_float4(1.f, 1.f, 1.f, 1.f);
            e1[b] = a.accumulate ? *reinterpret_cast<const float4*>(a.out + off) : make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
      }
    }

    f32x4 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (SPLIT) {
      // Step t = dy * KS + ks: A = the two planes of the Toeplitz fragment of PSF row dy, input columns 32 ks .. + 31;
      // B = 8 consecutive fp16 of window row 16 b + n + (kh - 1 - dy) at column 16 wave + 32 ks + 8 kk (one ds_read_b128
      // per plane).  The operands of step t + 1 are read while the 12 MFMAs of step t issue.
      const _Float16* bh = winH + (n + a.kh - 1) * PH + wave * 16 + 8 * kk;
      const _Float16* bl = winL + (n + a.kh - 1) * PH + wave * 16 + 8 * kk;
      const uint4* ap = afl16 + lane;
      const int n_steps = a.kh * KS;
      // (hi, hi) and (lo, hi) products first: their operands are double buffered; the lo plane of the window is read
      // at the start of a step into ONE buffer and is back before the third product needs it (8 MFMAs = 128 cycles
      // later) -- 16 registers less than double buffering it, which is what lets the epilogue's operands be
      // requested before the matrix phase without losing the second block per CU
      struct Ops {
        f16x8 ah, al, bh[4];
      };
      Ops o0, o1;
      f16x8 blv[4];
      f32x4 accx[4];  // the cross products (their lo operands carry a factor 2^11)
#pragma unroll
      for (int b = 0; b < 4; ++b) accx[b] = f32x4{0.f, 0.f, 0.f, 0.f};
      auto load_step = [&](Ops& o, int t) {
        t = t < n_steps ? t : n_steps - 1;
        const int dy = t / KS, ks = t - dy * KS;
        o.ah = __builtin_bit_cast(f16x8, ap[(t * 2) * 64]);
        o.al = __builtin_bit_cast(f16x8, ap[(t * 2 + 1) * 64]);
#pragma unroll
        for (int b = 0; b < 4; ++b) o.bh[b] = *reinterpret_cast<const f16x8*>(bh + (16 * b - dy) * PH + 32 * ks);
      };
      auto mfma_step = [&](const Ops& o, int t) {
        const int dy = t / KS, ks = t - dy * KS;
#pragma unroll
        for (int b = 0; b < 4; ++b) blv[b] = *reinterpret_cast<const f16x8*>(bl + (16 * b - dy) * PH + 32 * ks);
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(o.ah, o.bh[b], acc[b], 0, 0, 0);
#pragma unroll
        for (int b = 0; b < 4; ++b) accx[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(o.al, o.bh[b], accx[b], 0, 0, 0);
#pragma unroll
        for (int b = 0; b < 4; ++b) accx[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(o.ah, blv[b], accx[b], 0, 0, 0);
      };
      load_step(o0, 0);
      for (int t = 0; t < n_steps; t += 2) {
        load_step(o1, t + 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(o0, t);
        __builtin_amdgcn_sched_barrier(0);
        load_step(o0, t + 2);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < n_steps) mfma_step(o1, t + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[b] = (acc[b] + accx[b] * (1.f / 2048.f)) * inv;
    } else {
    // window row of output row (16 b + n) for PSF row dy is 16 b + n + (kh - 1 - dy).  The operands of
    // PSF row dy + 1 are read from LDS while the MFMAs of row dy issue (register double buffer, the
    // loop is unrolled by two so no copies are needed); reads past the last row are clamped.
    const float* bp = win + (n + a.kh - 1) * PITCH + wave * 16 + kk;
    const float* ap = afl + lane;
    float a0[STEPS], a1[STEPS], b0[STEPS][4], b1[STEPS][4];
    auto load_row = [&](float (&av)[STEPS], float (&bv)[STEPS][4], int dy) {
      const int d = dy < a.kh ? dy : a.kh - 1;
      const float* app = ap + d * (STEPS * 64);
      const float* bpp = bp - d * PITCH;
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        av[s] = app[s * 64];
#pragma unroll
        for (int b = 0; b < 4; ++b) bv[s][b] = bpp[b * 16 * PITCH + 4 * s];
      }
    };
    auto mfma_row = [&](const float (&av)[STEPS], const float (&bv)[STEPS][4]) {
#pragma unroll
      for (int s = 0; s < STEPS; ++s)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s][b], acc[b], 0, 0, 0);
    };
    load_row(a0, b0, 0);
    for (int dy = 0; dy < a.kh; dy += 2) {
      // sched_barrier: keep the LDS reads of the next row AHEAD of this row's MFMAs (the scheduler
      // otherwise sinks them to just before their use and exposes the LDS latency)
      load_row(a1, b1, dy + 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_row(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      load_row(a0, b0, dy + 2);
      __builtin_amdgcn_sched_barrier(0);
      if (dy + 1 < a.kh) mfma_row(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
    }

    }
    // ---- epilogue: lane holds out[y0 + 16 b + n][x0 + 16 wave + 4 kk + 0..3] ---------------------
    __shared__ double red[4];
    if constexpr (POISSON) {
      // the thread's sixteen loss terms are summed in fp32 before they join the fp64 sum of the tile (poisson_point:
      // the arithmetic every Poisson pass of the library shares)
      float local = 0.f;
      if (vec) {
        float4 b4[4], c4[4];
        bool live[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int y = y0 + 16 * b + n;
          live[b] = y < a.H;
          if constexpr (SPLIT) {
            b4[b] = e0[b], c4[b] = e1[b];
          } else {
            const size_t off = (size_t)(live[b] ? y : y0) * a.W + x;  // row y0 always exists
            b4[b] = *reinterpret_cast<const float4*>(a.background + off);
            c4[b] = *reinterpret_cast<const float4*>(a.counts + off);
          }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          if (!live[b]) continue;
          const size_t off = (size_t)(y0 + 16 * b + n) * a.W + x;
          const float bg[4] = {b4[b].x, b4[b].y, b4[b].z, b4[b].w}, cn[4] = {c4[b].x, c4[b].y, c4[b].z, c4[b].w};
          float np[4], g[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float term;
            np[i] = fmaxf(acc[b][i], 0.f) + bg[i];
            poisson_point(np[i], cn[i], a.eps, a.inv_n, term, g[i]);
            local += term;
            g[i] = acc[b][i] >= 0.f ? g[i] : 0.f;  // clamp backward: the gradient passes where conv >= 0
          }
          if (a.write_grad) *reinterpret_cast<float4*>(a.out + off) = make_float4(g[0], g[1], g[2], g[3]);
          if (a.npred_out) *reinterpret_cast<float4*>(a.npred_out + off) = make_float4(np[0], np[1], np[2], np[3]);
        }
      } else {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int y = y0 + 16 * b + n;
          if (y >= a.H || x >= a.W) continue;
          const size_t off = (size_t)y * a.W + x;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (x + i >= a.W) break;
            float term, g;
            const float np = fmaxf(acc[b][i], 0.f) + a.background[off + i];
            poisson_point(np, a.counts[off + i], a.eps, a.inv_n, term, g);
            local += term;
            if (a.write_grad) a.out[off + i] = acc[b][i] >= 0.f ? g : 0.f;
            if (a.npred_out) a.npred_out[off + i] = np;
          }
        }
      }
      const double wave_total = wave_sum((double)local);
      if (lane == 0) red[wave] = wave_total;
    } else if (vec) {
      // all loads of the four row groups are issued before the first one is consumed: one memory
      // round trip per tile instead of eight dependent ones
      float4 s4[4], o4[4];
      bool live[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int y = y0 + 16 * b + n;
        live[b] = y < a.H;
        if constexpr (SPLIT) {
          s4[b] = e0[b], o4[b] = e1[b];
        } else {
          const size_t off = (size_t)(live[b] ? y : y0) * a.W + x;  // row y0 always exists
          s4[b] = a.out_scale ? *reinterpret_cast<const float4*>(a.out_scale + off) : make_float4(1.f, 1.f, 1.f, 1.f);
          o4[b] = a.accumulate ? *reinterpret_cast<const float4*>(a.out + off) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (!live[b]) continue;
        const size_t off = (size_t)(y0 + 16 * b + n) * a.W + x;
        *reinterpret_cast<float4*>(a.out + off) =
            make_float4(fmaf(acc[b][0] * a.coef, s4[b].x, o4[b].x), fmaf(acc[b][1] * a.coef, s4[b].y, o4[b].y),
                        fmaf(acc[b][2] * a.coef, s4[b].z, o4[b].z), fmaf(acc[b][3] * a.coef, s4[b].w, o4[b].w));
      }
    } else {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int y = y0 + 16 * b + n;
        if (y >= a.H || x >= a.W) continue;
        const size_t off = (size_t)y * a.W + x;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (x + i >= a.W) break;
          const float sc = a.out_scale ? a.out_scale[off + i] : 1.f;
          const float ov = a.accumulate ? a.out[off + i] : 0.f;
          a.out[off + i] = fmaf(acc[b][i] * a.coef, sc, ov);
        }
      }
    }
    __syncthreads();  // every wave is done reading the window before it is overwritten
    // (the next write of `red` comes after the barrier at the top of the next tile)
    if (POISSON && threadIdx.x == 0) a.partials[tile] = (red[0] + red[1]) + (red[2] + red[3]);
    tile = next;
  }
}

// Toeplitz fragments of one PSF: afrag[dy][s][lane] = psf'[dy][m + kw - 1 - c], m = lane & 15,
// c = 4 s + (lane >> 4), psf' = psf (forward) or psf flipped in both axes (adjoint).
__global__ __launch_bounds__(256) void toeplitz_fragments_kernel(const float* __restrict__ psf, float* __restrict__ afrag,
                                                                int kh, int kw, int steps, int flip) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= kh * steps * 64) return;
  const int lane = i & 63, s = (i >> 6) % steps, dy = (i >> 6) / steps;
  const int m = lane & 15, c = 4 * s + (lane >> 4);
  const int dx = m + kw - 1 - c;
  float v = 0.f;
  if (dx >= 0 && dx < kw) v = flip ? psf[(kh - 1 - dy) * kw + (kw - 1 - dx)] : psf[dy * kw + dx];
  afrag[i] = v;
}

// SPLIT tables: afrag16[(dy * KS + ks) * 2 + plane][lane] = 8 fp16 of psf'[dy][m + kw - 1 - c] * s_p, m = lane & 15,
// c = 32 ks + 8 (lane >> 4) + e; plane 0 = fp16(v), plane 1 = fp16(2^11 (v - plane 0)); s_p = the power of two that puts
// max |psf| into [2^13, 2^14); the float after the table is 1 / s_p.  One block (a PSF has at most 33 x 33 taps).
__global__ __launch_bounds__(256) void toeplitz_fragments16_kernel(const float* __restrict__ psf, uint4* __restrict__ afrag,
                                                                  int kh, int kw, int ks_count, int flip) {
  __shared__ float red[4];
  float m = 0.f;
  for (int i = threadIdx.x; i < kh * kw; i += 256) m = fmaxf(m, fabsf(psf[i]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  int ex = 14;
  if (m > 0.f && m < 3.0e38f) (void)frexpf(m, &ex);
  ex = ex < -100 ? -100 : ex;
  const float sp = ldexpf(1.f, 14 - ex);
  for (int i = threadIdx.x; i < kh * ks_count * 64; i += 256) {
    const int lane = i & 63, t = i >> 6, ks = t % ks_count, dy = t / ks_count;
    const int mrow = lane & 15, g = lane >> 4;
    f16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = 32 * ks + 8 * g + e;
      const int dx = mrow + kw - 1 - c;
      float v = 0.f;
      if (dx >= 0 && dx < kw) v = flip ? psf[(kh - 1 - dy) * kw + (kw - 1 - dx)] : psf[dy * kw + dx];
      v *= sp;
      hi[e] = (_Float16)v;
      lo[e] = (_Float16)((v - (float)hi[e]) * 2048.f);
    }
    afrag[(t * 2) * 64 + lane] = __builtin_bit_cast(uint4, hi);
    afrag[(t * 2 + 1) * 64 + lane] = __builtin_bit_cast(uint4, lo);
  }
  if (threadIdx.x == 0) reinterpret_cast<float*>(afrag)[(size_t)kh * ks_count * 512] = ldexpf(1.f, ex - 14);
}

int direct_conv_kc(int kw) { return ((16 + kw - 1) + 3) / 4 * 4; }

bool direct_conv_supported(int kh, int kw) { return kh >= 1 && kw >= 1 && kh <= 33 && direct_conv_kc(kw) <= 48; }

static int direct_split_ks(int kw) { return (16 + kw - 1 + 31) / 32; }
static size_t direct_split_lds(int kh, int kw) {
  const int ks = direct_split_ks(kw), rows = TILE - 1 + kh, ph = 48 + 32 * ks + 8;
  return (size_t)(((rows * ph + 3) & ~3)) * sizeof(float) + (size_t)kh * ks * 128 * sizeof(uint4);
}
// the split-fp16 kernel needs its two window planes and its fragment table in the 160 KB of LDS
bool direct_conv_split_supported(int kh, int kw) {
  return direct_conv_supported(kh, kw) && direct_split_ks(kw) <= 2 && direct_split_lds(kh, kw) <= 160 * 1024;
}

size_t direct_conv_fragment_floats(int kh, int kw, int split) {
  if (split) return (size_t)kh * direct_split_ks(kw) * 512 + 4;
  return (size_t)kh * (direct_conv_kc(kw) / 4) * 64;
}

int launch_toeplitz_fragments(const float* psf, float* afrag_fwd, float* afrag_adj, int kh, int kw, int split,
                              hipStream_t stream) {
  if (split) {
    const int ks = direct_split_ks(kw);
    toeplitz_fragments16_kernel<<<1, 256, 0, stream>>>(psf, reinterpret_cast<uint4*>(afrag_fwd), kh, kw, ks, 0);
    toeplitz_fragments16_kernel<<<1, 256, 0, stream>>>(psf, reinterpret_cast<uint4*>(afrag_adj), kh, kw, ks, 1);
    JD_LAUNCH_CHECK();
    return JD_OK;
  }
  const int steps = direct_conv_kc(kw) / 4;
  const int n = kh * steps * 64;
  toeplitz_fragments_kernel<<<(n + 255) / 256, 256, 0, stream>>>(psf, afrag_fwd, kh, kw, steps, 0);
  toeplitz_fragments_kernel<<<(n + 255) / 256, 256, 0, stream>>>(psf, afrag_adj, kh, kw, steps, 1);
  JD_LAUNCH_CHECK();
  return JD_OK;
}

static int g_conv_n_cu = 0;

template <int KC, bool POISSON, bool SPLIT = false>
static int launch_kc(const DirectConvArgs& a, hipStream_t stream) {
  constexpr int PITCH = 48 + KC + 2, PH = 48 + KC + 8;
  const int rows = TILE - 1 + a.kh;
  const size_t lds = SPLIT ? (size_t)((rows * PH + 3) & ~3) * sizeof(float) + (size_t)a.kh * (KC / 32) * 128 * sizeof(uint4)
                           : (size_t)(((rows * PITCH + 3) & ~3) + a.kh * (KC / 4) * 64) * sizeof(float);
  static int configured_bytes = 0;
  if ((int)lds > configured_bytes) {
    JD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&direct_conv_kernel<KC, true, POISSON, SPLIT>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    JD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&direct_conv_kernel<KC, false, POISSON, SPLIT>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    configured_bytes = (int)lds;
  }
  if (!g_conv_n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    g_conv_n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
                      ? prop.multiProcessorCount
                      : 256;
  }
  const int n_tiles = ((a.W + TILE - 1) / TILE) * ((a.H + TILE - 1) / TILE);
  // persistent grid: as many blocks as fit at once (LDS bound, at most 2 per CU: 2 waves per SIMD)
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu > 2) per_cu = 2;
  if (per_cu < 1) per_cu = 1;
  {  // tuning override
    const int v = opt_value(OPT_CONV_BLOCKS_PER_CU, 0);
    if (v >= 1 && v <= 8) per_cu = v;
  }
  int grid = g_conv_n_cu * per_cu;
  if (grid > n_tiles) grid = n_tiles;
  const bool vec = (a.W % 4 == 0) && (a.ox_in % 4 == 0) &&
                   (reinterpret_cast<uintptr_t>(a.in) % 16 == 0) &&
                   (!a.in_scale || reinterpret_cast<uintptr_t>(a.in_scale) % 16 == 0);
  if (vec)
    direct_conv_kernel<KC, true, POISSON, SPLIT><<<grid, 256, lds, stream>>>(a);
  else
    direct_conv_kernel<KC, false, POISSON, SPLIT><<<grid, 256, lds, stream>>>(a);
  JD_LAUNCH_CHECK();
  return JD_OK;
}

template <bool POISSON>
static int dispatch_kc(const DirectConvArgs& a, int kw, int split, hipStream_t stream) {
  if (split) return direct_split_ks(kw) == 1 ? launch_kc<32, POISSON, true>(a, stream) : launch_kc<64, POISSON, true>(a, stream);
  switch (direct_conv_kc(kw)) {
    case 16: return launch_kc<16, POISSON>(a, stream);
    case 20: return launch_kc<20, POISSON>(a, stream);
    case 24: return launch_kc<24, POISSON>(a, stream);
    case 28: return launch_kc<28, POISSON>(a, stream);
    case 32: return launch_kc<32, POISSON>(a, stream);
    case 36: return launch_kc<36, POISSON>(a, stream);
    case 40: return launch_kc<40, POISSON>(a, stream);
    case 44: return launch_kc<44, POISSON>(a, stream);
    case 48: return launch_kc<48, POISSON>(a, stream);
    default: return fail(JD_ERR_INVALID, "direct convolution: PSF width %d not supported", kw);
  }
}

// adjoint == 0: out (+)= coef * out_scale * conv_same(in * in_scale, psf)    [crop offset (oy, ox)]
// adjoint != 0: out (+)= coef * out_scale * corr_same(in * in_scale, psf)    (the transpose of the above)
int launch_direct_conv(const float* in, const float* in_scale, const float* afrag, float* out, const float* out_scale,
                       int H, int W, int kh, int kw, int oy, int ox, int adjoint, float coef, int accumulate, int split,
                       hipStream_t stream) {
  DirectConvArgs a{};
  a.in = in, a.in_scale = in_scale, a.afrag = afrag, a.out = out, a.out_scale = out_scale;
  a.H = H, a.W = W, a.kh = kh, a.coef = coef, a.accumulate = accumulate;
  if (adjoint) {
    a.oy = kh - 1 - oy;
    a.ox_in = -ox;  // (kw - 1 - ox) - (kw - 1)
  } else {
    a.oy = oy;
    a.ox_in = ox - (kw - 1);
  }
  ProfScope prof(JD_KERNEL_DIRECT_CONV, stream);
  return dispatch_kc<false>(a, kw, split, stream);
}

int direct_conv_tiles(int H, int W) { return ((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE); }

// Forward model of one component with the Poisson pass as the epilogue:
//   n = max(conv_same(in * in_scale, psf), 0) + background;  partials[tile] = sum of n - counts log(n + eps);
//   g_out = (1 - counts / (n + eps)) * inv_n where conv >= 0, else 0   (only if write_grad);  npred_out = n (nullable)
int launch_direct_conv_poisson(const float* in, const float* in_scale, const float* afrag, float* g_out, int H, int W,
                               int kh, int kw, int oy, int ox, const float* background, const float* counts,
                               float* npred_out, double* partials, float eps, float inv_n, int write_grad,
                               int* n_partials, int split, hipStream_t stream) {
  DirectConvArgs a{};
  a.in = in, a.in_scale = in_scale, a.afrag = afrag, a.out = g_out;
  a.H = H, a.W = W, a.kh = kh, a.coef = 1.f;
  a.oy = oy, a.ox_in = ox - (kw - 1);
  a.background = background, a.counts = counts, a.npred_out = npred_out, a.partials = partials;
  a.eps = eps, a.inv_n = inv_n, a.write_grad = write_grad;
  *n_partials = direct_conv_tiles(H, W);
  ProfScope prof(JD_KERNEL_POISSON_FUSED, stream);  // the fused launch IS the Poisson pass
  return dispatch_kc<true>(a, kw, split, stream);
}

}  // namespace jd
