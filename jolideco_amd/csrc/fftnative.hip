// Native FFT convolution: the 'same' convolution of jolideco/utils/torch.py:347-370 (rfft2 -> k-space product -> irfft2 ->
// centre crop) as THREE launches of hand-written CDNA4 kernels instead of rocFFT's ~12 (2 x 6) + pad + product + crop:
//
//   rows    one block per PAIR of image rows: z = (a * scale)[y] + i (a * scale)[y + H/2] -- the image is split into its
//           upper and lower half, one in the real and one in the imaginary part; a convolution with a REAL kernel acts on
//           both independently -- zero padded to Nx, complex FFT of length Nx in LDS, spectrum row to HBM.  No padded
//           real image, no Hermitian bookkeeping: every transform of the path is a plain complex FFT.  With a
//           calibration the input is the bilinearly shifted image (the shift kernel's arithmetic inside the load).
//   columns one wave per column (four columns per block; two waves per column for 4096-row images): the H/2 spectrum rows
//           of the column, zero extended to Ny, in LDS and in place: the forward passes but the last; then the last
//           forward pass, the product with the kernel spectrum K^ (column-major) and the first inverse pass in one
//           register block (the inverse transform's radices run in reverse order: its first butterfly reads what the
//           forward's last one holds); then the remaining inverse passes.
//   rows^-1 one block per row pair: inverse FFT of length Nx, then the epilogue on the un-padded image: Re -> row y, Im ->
//           row y + H/2.  Where the convolution of one half spills into rows of the other (the halves are convolved as
//           separate images), the needed part of the spill row is added in the Fourier domain BEFORE the transform
//           (load_spectrum_row): every block runs one transform.  Epilogues: plain store, or the adjoint's
//           grad (+)= coef * exposure * corr (+ the loss of the dataset, finalised by block 0).
//   A dataset's likelihood step is FIVE launches: rows, columns, [rows^-1 + Poisson pass + rows of g] (with up-sampling:
//   U inverse transforms + sum-pool + Poisson pass + one transform of the up-sampled g rows), columns, rows^-1 + adjoint
//   epilogue; for the datasets of a joint step every launch covers all of them (FftBatch).
//
// Row and column lengths of 1024- / 2048- / 4096-pixel images run kernels whose radix schedule, length and loop bounds are
// template parameters; the generic runtime-radix kernels serve every other length 2^a * {1, 3, 9} (jd_fftcore.h), rocFFT
// everything else.
//
// HBM traffic of one forward convolution at 2048^2 with a 33 x 33 PSF (Nx = 2304, Ny = 1152): 33.6 MB in, 18.9 MB
// spectrum out, 18.9 + 21.2 (K^) in, 21.2 out, 21.2 in, 16.8 out = 152 MB against rocFFT's ~6 passes over a 2100 x 2100
// grid per transform (profiles/r01).
#include <algorithm>
#include <cmath>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "jd_common.h"
#include "jd_fftcore.h"
#include "kernels.h"

namespace jd {

namespace {

using namespace jdfft;

constexpr int ROW_THREADS = 256;

struct FftPasses {
  int n, r[MAX_PASSES];
};

__device__ __forceinline__ void lds_wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

#ifndef JD_FFT_LOOP_OUTSIDE
#define JD_FFT_LOOP_OUTSIDE 0  // 1: one butterfly loop per radix (switch outside): more registers, fewer instructions
#endif
// (measured and removed, round 5: a register prefetch of the pooled launch's next spectrum row while the block transforms the
// current one -- c6: 68 -> 88 us per launch, 96 registers are not enough for it; the rows are summed before ONE transform now)
#ifndef JD_FFT_LIN
#define JD_FFT_LIN 0           // 1: linear padded indices where a pass allows them (jd_fftcore.h)
#endif

// one pass of radix R by `nthreads` threads: butterflies tid, tid + nthreads, ...
template <int R, int DIR>
__device__ __forceinline__ void pass_loop(const float2* x, float2* y, int N, int p, const float2* tw, int tid, int nthreads) {
  if (JD_FFT_LIN && pass_is_linear(N, R, p)) {  // (uniform)
    for (int i = tid; i < N / R; i += nthreads) pass_one<R, DIR, true>(x, y, N, p, tw, i);
  } else {
    for (int i = tid; i < N / R; i += nthreads) pass_one<R, DIR, false>(x, y, N, p, tw, i);
  }
}

template <int DIR>
__device__ __forceinline__ void pass_dispatch(int R, const float2* x, float2* y, int N, int p, const float2* tw, int b) {
  switch (R) {
    case 16: pass_one<16, DIR>(x, y, N, p, tw, b); break;
    case 8: pass_one<8, DIR>(x, y, N, p, tw, b); break;
    case 4: pass_one<4, DIR>(x, y, N, p, tw, b); break;
    case 2: pass_one<2, DIR>(x, y, N, p, tw, b); break;
    case 9: pass_one<9, DIR>(x, y, N, p, tw, b); break;
    default: pass_one<3, DIR>(x, y, N, p, tw, b); break;
  }
}

// FFT of one sequence by `nthreads` threads (a block: BLOCK_SYNC, or one wave); the result is in the returned buffer
template <int DIR, bool BLOCK_SYNC>
__device__ __forceinline__ float2* fft_lds(float2* a, float2* b, int N, const FftPasses& f, const float2* tw, int tid, int nthreads) {
  int p = 1;
  for (int s = 0; s < f.n; ++s) {
    const int R = f.r[s];
#if JD_FFT_LOOP_OUTSIDE
    switch (R) {  // (the loop over the butterflies INSIDE each case: one simple loop per radix)
      case 16: pass_loop<16, DIR>(a, b, N, p, tw, tid, nthreads); break;
      case 8: pass_loop<8, DIR>(a, b, N, p, tw, tid, nthreads); break;
      case 4: pass_loop<4, DIR>(a, b, N, p, tw, tid, nthreads); break;
      case 2: pass_loop<2, DIR>(a, b, N, p, tw, tid, nthreads); break;
      case 9: pass_loop<9, DIR>(a, b, N, p, tw, tid, nthreads); break;
      default: pass_loop<3, DIR>(a, b, N, p, tw, tid, nthreads); break;
    }
#else
    for (int i = tid; i < N / R; i += nthreads) pass_dispatch<DIR>(R, a, b, N, p, tw, i);
#endif
    if (BLOCK_SYNC) __syncthreads();
    else lds_wave_fence();
    p *= R;
    float2* t = a;
    a = b, b = t;
  }
  return a;
}

// Compile-time radix schedule (R0, R1, R2[, R3]) of a row transform: N, every sub-transform length and every loop bound
// are constants (the generic form dispatches on the radix inside the butterfly loop).  R0 = 0: generic.
// T: threads of a row block.  4608 = 8 * 8 * 8 * 9 runs 576 threads (nine waves): ONE radix-8 butterfly per thread and pass
// (576 of them; 512 radix-9 butterflies), and two blocks -- LDS holds no more: two padded sequences of 4608 are 78 KB --
// are 18 waves per CU instead of 8.  MAXQ / PRE: 16-byte pieces of an image row / of a spectrum row per thread.
// The generic (runtime-radix) form comes in two sizes: <0, 0, 0, 1> for rows of at most GENERIC_SMALL points (image and padded
// length: the per-thread arrays of a row's pieces are a third as long -- the small images of the reference's own examples)
// and <0, 0, 0, 0> for everything else.
// <0, 0, 0, 2>: rows of at most GENERIC_TINY points on ONE wave (round 5: a 768-point row is 48 radix-16 butterflies per pass --
// a 256-thread block idles through them and pays block barriers between the passes; one wave per row pair fences instead).
constexpr int GENERIC_TINY = 1024, GENERIC_SMALL = 1536, GENERIC_LARGE = 5120;
template <int R0, int R1, int R2, int R3>
struct RowSched {
  static constexpr bool STATIC = R0 > 0;
  static constexpr int N = STATIC ? R0 * R1 * R2 * (R3 ? R3 : 1) : 0;
  static constexpr int LIMIT = STATIC ? N : R3 == 2 ? GENERIC_TINY : R3 == 1 ? GENERIC_SMALL : GENERIC_LARGE;  // longest row (pixels or points) the kernel takes
  static constexpr int T = (R0 == 8 && R1 == 8 && R2 == 8 && R3 == 9) ? 576 : (!STATIC && R3 == 2) ? 64 : ROW_THREADS;
  static constexpr int WAVES_PER_SIMD = T == 576 ? 5 : 1;  // (two blocks of nine waves: five on one SIMD -> at most 96 registers)
  static constexpr int MAXQ = (LIMIT + 4 * T - 1) / (4 * T);  // 16-byte pieces of an image row per thread
  static constexpr int PRE = (LIMIT + 2 * T - 1) / (2 * T);   // 16-byte pieces of a spectrum row per thread
};

template <int R, int DIR, int N, int P, int T>
__device__ __forceinline__ void pass_static(const float2* x, float2* y, const float2* tw, int tid) {
#pragma unroll
  for (int i0 = 0; i0 < N / R; i0 += T) {
    const int i = i0 + tid;
    // linear padded indices where the pass allows them and a thread runs ONE butterfly per pass (576 threads): with
    // several, the compiler keeps them all in flight and the registers cost more waves than the instructions save
    constexpr bool LIN = (T == 576 || JD_FFT_LIN) && pass_is_linear(N, R, P);
    if (i < N / R) pass_one<R, DIR, LIN>(x, y, N, P, tw, i);
  }
  __syncthreads();
}

template <int DIR, int R0, int R1, int R2, int R3>
__device__ __forceinline__ float2* fft_lds_static(float2* a, float2* b, const float2* tw, int tid) {
  constexpr int N = RowSched<R0, R1, R2, R3>::N, T = RowSched<R0, R1, R2, R3>::T;
  pass_static<R0, DIR, N, 1, T>(a, b, tw, tid);
  pass_static<R1, DIR, N, R0, T>(b, a, tw, tid);
  pass_static<R2, DIR, N, R0 * R1, T>(a, b, tw, tid);
  if constexpr (R3 > 0) {
    pass_static<R3, DIR, N, R0 * R1 * R2, T>(b, a, tw, tid);
    return a;
  } else {
    return b;
  }
}

// the row transform of a kernel instantiated for the schedule S (generic: the passes of `f`)
template <int DIR, class S, int R0, int R1, int R2, int R3>
__device__ __forceinline__ float2* row_fft(float2* a, float2* b, int Nx, const FftPasses& f, const float2* tw, int tid) {
  if constexpr (S::STATIC) return fft_lds_static<DIR, R0, R1, R2, R3>(a, b, tw, tid);
  else return fft_lds<DIR, (S::T > 64)>(a, b, Nx, f, tw, tid, S::T);
}

// The same transform IN PLACE by the LANES threads of one column (the column kernel: one buffer per column instead of
// two, twice the columns in flight per CU): in every pass a thread first reads the inputs of all its butterflies (at
// most MAXB of them) into registers, then writes their outputs.  LANES = 64: one wave per column -- LDS operations of one
// wave execute in program order, so all reads of a pass are done before its first write lands, and the fence only keeps
// the compiler from reordering them.  LANES = 128: two waves per column (the long columns of 4096-row images: half the
// butterflies per thread, no register spills), block barriers between the phases -- every column of the block runs the
// same pass schedule.
template <int LANES>
__device__ __forceinline__ void column_sync() {
  if constexpr (LANES == 64) lds_wave_fence();
  else __syncthreads();
}

// k = b mod p for b < 2^23 and any p >= 1 (the sub-transform lengths of the inverse column transform are not powers of
// two: its radices run in reverse order, the odd one first)
__device__ __forceinline__ int mod_small(int b, int p, float inv_p) {
  int q = (int)(((float)b + 0.5f) * inv_p);
  int k = b - q * p;
  k += k < 0 ? p : 0;
  return k >= p ? k - p : k;
}

// LIN_IN / LIN_OUT (compile-time schedules only): the padded index is linear in t -- inputs when N / R is a multiple of 16
// (lp(b + t nb) = lp(b) + t (nb + nb / 16)), outputs when p is a multiple of 16, or p = 1 and the R <= 16 outputs of a
// butterfly share one group of 16 (R = 2, 4, 8, 16) -- one address per butterfly and compile-time offsets in the LDS
// instructions instead of an add, a shift and an add per element (a third of the kernel's vector instructions).
constexpr bool lin_in(int N, int R) { return ((N / R) & 15) == 0; }
constexpr bool lin_out(int R, int p) { return (p & 15) == 0 || (p == 1 && (R == 2 || R == 4 || R == 8 || R == 16)); }

template <int R, int DIR, int MAXB, int LANES, bool POW2, bool LIN_IN = false, bool LIN_OUT = false>
__device__ __forceinline__ void column_pass_inplace(float2* x_, int N, int p, const float2* tw_, int lane) {
  cf* x = reinterpret_cast<cf*>(x_);
  const cf* tw = reinterpret_cast<const cf*>(tw_);
  const int nb = N / R;
  const float inv_p = 1.f / (float)p;
  cf u[MAXB][R];
#pragma unroll
  for (int q = 0; q < MAXB; ++q) {
    const int b = lane + LANES * q;
    if (b < nb) {
      if constexpr (LIN_IN) {
        const cf* xb = x + lp(b);
        const int step = nb + (nb >> 4);
#pragma unroll
        for (int t = 0; t < R; ++t) u[q][t] = xb[t * step];
      } else {
#pragma unroll
        for (int t = 0; t < R; ++t) u[q][t] = x[lp(b + t * nb)];
      }
    }
  }
  column_sync<LANES>();
#pragma unroll
  for (int q = 0; q < MAXB; ++q) {
    const int b = lane + LANES * q;
    if (b < nb) {
      const int k = POW2 ? (b & (p - 1)) : mod_small(b, p, inv_p);
      if (p > 1) {
        cf w[R];
        w[1] = tw[k * (nb / p)];
        if (DIR > 0) w[1].y = -w[1].y;
#pragma unroll
        for (int t = 2; t < R; ++t) w[t] = cmul(w[t / 2], w[t - t / 2]);
#pragma unroll
        for (int t = 1; t < R; ++t) u[q][t] = cmul(u[q][t], w[t]);
      }
      Dft<R, DIR>::run(u[q]);
      const int j = (b - k) * R + k;
      if constexpr (LIN_OUT) {
        cf* xj = x + lp(j);
        const int step = p == 1 ? 1 : p + (p >> 4);
#pragma unroll
        for (int t = 0; t < R; ++t) xj[t * step] = u[q][t];
      } else {
#pragma unroll
        for (int t = 0; t < R; ++t) x[lp(j + t * p)] = u[q][t];
      }
    }
  }
  column_sync<LANES>();
}

// The middle of the column pass in registers: the LAST pass of the forward transform (radix R, sub-transform length
// p = N / R: butterfly b reads x[b + t p] and produces the spectrum elements b + t p), the product with the kernel
// spectrum, and the FIRST pass of the inverse transform, whose radices run in reverse order -- its first pass (radix R,
// sub-transform length 1) reads exactly the elements b + t p the thread holds.  Two LDS exchanges and the separate pass
// over the column for the product are gone.
template <int R, int MAXB, int LANES, bool LIN_IN = false>
__device__ __forceinline__ void column_mid_inplace(float2* x_, int N, const float2* tw_, const float2* kcol_, int conj, int lane) {
  cf* x = reinterpret_cast<cf*>(x_);
  const cf* tw = reinterpret_cast<const cf*>(tw_);
  const cf* kcol = reinterpret_cast<const cf*>(kcol_);
  const int nb = N / R;  // = p of the forward pass
  cf u[MAXB][R];
#pragma unroll
  for (int q = 0; q < MAXB; ++q) {
    const int b = lane + LANES * q;
    if (b < nb) {
      if constexpr (LIN_IN) {
        const cf* xb = x + lp(b);
        const int step = nb + (nb >> 4);
#pragma unroll
        for (int t = 0; t < R; ++t) u[q][t] = xb[t * step];
      } else {
#pragma unroll
        for (int t = 0; t < R; ++t) u[q][t] = x[lp(b + t * nb)];
      }
    }
  }
  column_sync<LANES>();
#pragma unroll
  for (int q = 0; q < MAXB; ++q) {
    const int b = lane + LANES * q;
    if (b < nb) {
      cf kh[R];
#pragma unroll
      for (int t = 0; t < R; ++t) {
#if defined(__HIP_DEVICE_COMPILE__)  // (address space 1: the table's pointer is generic; cf is a builtin vector on the device only)
        kh[t] = *((const JD_AS1 cf*)kcol + (b + t * nb));
#else
        kh[t] = kcol[b + t * nb];
#endif
      }
      {
        cf w[R];
        w[1] = tw[b];  // k = b, nb / p = 1
#pragma unroll
        for (int t = 2; t < R; ++t) w[t] = cmul(w[t / 2], w[t - t / 2]);
#pragma unroll
        for (int t = 1; t < R; ++t) u[q][t] = cmul(u[q][t], w[t]);
      }
      Dft<R, -1>::run(u[q]);
#pragma unroll
      for (int t = 0; t < R; ++t) {
        cf k = kh[t];
        if (conj) k.y = -k.y;
        u[q][t] = cmul(u[q][t], k);
      }
      Dft<R, 1>::run(u[q]);
      if constexpr (LIN_IN && lin_out(R, 1)) {  // (compile-time schedules; the R outputs share a group of 16)
        cf* xj = x + lp(b * R);
#pragma unroll
        for (int t = 0; t < R; ++t) xj[t] = u[q][t];
      } else {
#pragma unroll
        for (int t = 0; t < R; ++t) x[lp(b * R + t)] = u[q][t];
      }
    }
  }
  column_sync<LANES>();
}

// butterflies per thread (N / R / LANES rounded up) the instantiations hold: 2 / 3 / 2 / 2 at radix 16 / 8 / 9 / 4 --
// one wave per column up to N = 1152 (16: 72 butterflies, 8: 144, 9: 128) and N = 1024, two waves up to 2304 (144 / 256 /
// 256) and 2048; 0: the length is not a column length (radix 2 or 3 passes, a single pass, or too long)
__host__ __device__ inline int column_lanes(int N, const FftPasses& f) {
  if (f.n < 2) return 0;
  for (int lanes = 64; lanes <= 128; lanes *= 2) {
    bool ok = true;
    for (int s = 0; s < f.n; ++s) {
      const int R = f.r[s], per = (N / R + lanes - 1) / lanes;
      ok = ok && ((R == 16 && per <= 2) || (R == 8 && per <= 3) || (R == 9 && per <= 2) || (R == 4 && per <= 2));
    }
    if (ok) return lanes;
  }
  return 0;
}

template <int DIR, int LANES, bool POW2>
__device__ __forceinline__ void column_pass_any(int R, float2* x, int N, int p, const float2* tw, int lane) {
  switch (R) {
    case 16: column_pass_inplace<16, DIR, 2, LANES, POW2>(x, N, p, tw, lane); break;
    case 8: column_pass_inplace<8, DIR, 3, LANES, POW2>(x, N, p, tw, lane); break;
    case 9: column_pass_inplace<9, DIR, 2, LANES, POW2>(x, N, p, tw, lane); break;
    default: column_pass_inplace<4, DIR, 2, LANES, POW2>(x, N, p, tw, lane); break;
  }
}

// circular convolution of one column with the kernel whose spectrum is kcol: forward passes 0 .. n - 2, the fused middle
// (pass n - 1, product, first inverse pass), inverse passes n - 2 .. 0 (radices in reverse order; unnormalised, the
// normalisation is folded into the kernel spectrum)
template <int LANES>
__device__ __forceinline__ void column_conv_inplace(float2* x, int N, const FftPasses& f, const float2* tw, const float2* kcol,
                                                    int conj, int lane) {
  int p = 1;
  for (int s = 0; s + 1 < f.n; ++s) {
    column_pass_any<-1, LANES, true>(f.r[s], x, N, p, tw, lane);
    p *= f.r[s];
  }
  const int Rm = f.r[f.n - 1];
  switch (Rm) {
    case 16: column_mid_inplace<16, 2, LANES>(x, N, tw, kcol, conj, lane); break;
    case 8: column_mid_inplace<8, 3, LANES>(x, N, tw, kcol, conj, lane); break;
    case 9: column_mid_inplace<9, 2, LANES>(x, N, tw, kcol, conj, lane); break;
    default: column_mid_inplace<4, 2, LANES>(x, N, tw, kcol, conj, lane); break;
  }
  p = Rm;
  for (int s = f.n - 2; s >= 0; --s) {
    if (Rm == 9) column_pass_any<1, LANES, false>(f.r[s], x, N, p, tw, lane);
    else column_pass_any<1, LANES, true>(f.r[s], x, N, p, tw, lane);
    p *= f.r[s];
  }
}

// Per-dataset pointers of a batched likelihood step (several datasets of one flux image in every launch: the blocks of a
// launch then run in several rounds and at different stages, and the load, transform and store phases of the launch
// overlap -- a single dataset's launch is ONE round of blocks that all load, then all transform, then all store).
// n_batch = 0: one dataset, described by the scalar members of the kernel's arguments.  The table (struct FftBatch,
// kernels.h) lives in device memory: as a by-value kernel argument, indexing it with the dataset number made the compiler
// copy it to scratch in every kernel.

// Four consecutive pixels x .. x + 3 (x % 4 == 0, x < W) of an image row.  Rows of W % 4 == 0 pixels start 16-byte aligned: one
// 16-byte access.  RAGGED rows (any other W; round 5: every image size takes the native path) are read and written at 4-byte
// alignment, the last piece of a row element by element; pixels beyond W read as zero and are not written.  `live` false:
// the row does not exist (the lower half of an image with an odd number of rows is one row short) -- zeros, no store.
__device__ __forceinline__ float4 row_ld4(const float* row, int x, int W, bool ragged, bool live = true) {
  if (!live) return make_float4(0.f, 0.f, 0.f, 0.f);
  if (!ragged) return gld4(row + x);  // (global-memory accessors of jd_common.h: no flat loads through table pointers)
  if (x + 4 <= W) return gld4u(row + x);
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (x < W) r.x = gld(row + x);
  if (x + 1 < W) r.y = gld(row + x + 1);
  if (x + 2 < W) r.z = gld(row + x + 2);
  return r;
}

__device__ __forceinline__ void row_st4(float* row, int x, int W, bool ragged, float4 v, bool live = true) {
  if (!live) return;
  if (!ragged) {
    gst4(row + x, v);
  } else if (x + 4 <= W) {
    gst4u(row + x, v);
  } else {
    if (x < W) gst(row + x, v.x);
    if (x + 1 < W) gst(row + x + 1, v.y);
    if (x + 2 < W) gst(row + x + 2, v.z);
  }
}

struct RowsFwdArgs {
  const float* in;
  const float* shift_xy;  // nullable, device [2]: the input is the bilinearly shifted image (the calibration's shift_fwd)
  float shift_scale;
  const FftBatch* batch;  // device memory, nullable; n_batch > 0: block b = row pair b / n of dataset d0 + b % n (scale, spec from the table)
  int n_batch, d0;
  const float* scale;  // nullable
  float2* spec;        // [Hh][Nx]
  const float2* tw;
  int H, W, Hh, Nx;
  FftPasses f;
};

// rows: z[x] = in[y][x] s[y][x] + i in[y + Hh][x] s[y + Hh][x], zero beyond W -> FFT -> spec[y][.]
template <int R0, int R1, int R2, int R3>
__global__ __launch_bounds__((RowSched<R0, R1, R2, R3>::T), (RowSched<R0, R1, R2, R3>::WAVES_PER_SIMD)) void fftn_rows_fwd_kernel(RowsFwdArgs a) {
  extern __shared__ float2 lds[];
  using S = RowSched<R0, R1, R2, R3>;
  const int Nx = S::STATIC ? S::N : a.Nx;
  const int tid = threadIdx.x, nb = a.n_batch;
  const int y = nb ? (int)blockIdx.x / nb : (int)blockIdx.x, d = nb ? a.d0 + (int)blockIdx.x - y * nb : 0;
  const float* const scale = nb ? a.batch->exposure[d] : a.scale;
  float2* const spec = nb ? a.batch->spec[d] : a.spec;
  const float* const shift_xy = nb ? a.batch->shift_xy[d] : a.shift_xy;
  float2* bufa = lds;
  float2* bufb = lds + lp_size(Nx);
  const size_t ra = (size_t)y * a.W, rb = (size_t)(y + a.Hh) * a.W;
  const bool ragged = (a.W & 3) != 0, lower = y + a.Hh < a.H;  // (odd H: the lower half is one row short)
  constexpr int MAXQ = S::MAXQ;
  float pu[MAXQ][4], pv[MAXQ][4];  // this thread's pieces of the two (shifted) input rows; zero beyond W
#pragma unroll
  for (int q = 0; q < MAXQ; ++q)
#pragma unroll
    for (int i = 0; i < 4; ++i) pu[q][i] = pv[q][i] = 0.f;
  if (shift_xy) {  // (block-uniform)
    // the bilinearly shifted rows of both halves, straight from global memory: per piece the five source columns of the two
    // source rows (one 16-byte load at a 4-byte aligned address + one float each; element-wise with bounds checks where
    // the window leaves the image) -- the arithmetic of shift_fwd_kernel, whose launch and image this replaces
    const ShiftGeom g = shift_geom_of(cld(shift_xy), cld(shift_xy + 1), a.shift_scale);
    const float w00 = g.wx0 * g.wy0, w10 = g.wx1 * g.wy0, w01 = g.wx0 * g.wy1, w11 = g.wx1 * g.wy1;
    // every source row piece by ONE unconditional 16-byte load + one float at clamped coordinates (issue_row5, jd_common.h),
    // the pieces of a chunk -- two halves x two source rows x QB pieces, and their exposure -- in flight together, the zeros
    // of the outside put in afterwards in registers.  (Loads under the window's bounds tests ran one after the other, each
    // behind a full wait.)  QB = 2 pieces per chunk, 1 for the one-wave rows of small fits (four pieces per thread there).
    constexpr int QB = (MAXQ < 2 || S::T <= 64) ? 1 : 2;
#pragma unroll
    for (int q0 = 0; q0 < MAXQ; q0 += QB) {
      Raw5 raw[2][2][QB];
      float4 su[QB], sv[QB];
#pragma unroll
      for (int j = 0; j < QB; ++j) {
        const int q = q0 + j < MAXQ ? q0 + j : MAXQ - 1;
        const int x = 4 * (tid + q * S::T), xs = x < a.W ? x : 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int r0 = y + h * a.Hh + g.fy;
          raw[h][0][j] = issue_row5(a.in, a.H, a.W, r0, xs + g.fx);
          raw[h][1][j] = issue_row5(a.in, a.H, a.W, r0 + 1, xs + g.fx);
        }
        su[j] = sv[j] = make_float4(1.f, 1.f, 1.f, 1.f);
        if (scale) su[j] = row_ld4(scale + ra, xs, a.W, ragged), sv[j] = row_ld4(scale + rb, xs, a.W, ragged, lower);
      }
#pragma unroll
      for (int j = 0; j < QB; ++j) {
        const int q = q0 + j;
        if (q >= MAXQ) continue;
        const int x = 4 * (tid + q * S::T);
        const float se[2][4] = {{su[j].x, su[j].y, su[j].z, su[j].w}, {sv[j].x, sv[j].y, sv[j].z, sv[j].w}};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const Row5 t0 = finish_row5(raw[h][0][j]), t1 = finish_row5(raw[h][1][j]);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float v = x < a.W ? t0.v[i] * w00 + t0.v[i + 1] * w10 + t1.v[i] * w01 + t1.v[i + 1] * w11 : 0.f;
            (h ? pv : pu)[q][i] = scale ? v * se[h][i] : v;  // (the product of the loop below, made here)
          }
        }
      }
    }
    if (ragged || !lower) {  // (the interpolation of neighbours inside the image gives non-zero values for pixels outside it)
#pragma unroll
      for (int q = 0; q < MAXQ; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool inside = 4 * (tid + q * S::T) + i < a.W;
          pu[q][i] = inside ? pu[q][i] : 0.f, pv[q][i] = inside && lower ? pv[q][i] : 0.f;
        }
    }
  } else {
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
      const int x = 4 * (tid + q * S::T);
      if (x < a.W) {
        const float4 u = row_ld4(a.in + ra, x, a.W, ragged), v = row_ld4(a.in + rb, x, a.W, ragged, lower);
        pu[q][0] = u.x, pu[q][1] = u.y, pu[q][2] = u.z, pu[q][3] = u.w;
        pv[q][0] = v.x, pv[q][1] = v.y, pv[q][2] = v.z, pv[q][3] = v.w;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < MAXQ; ++q) {
    const int x = 4 * (tid + q * S::T);
    if (x >= Nx) continue;
    if (scale && x < a.W && !shift_xy) {  // (shifted rows: the exposure arrived with the pieces and is applied already)
      const float4 su = row_ld4(scale + ra, x, a.W, ragged), sv = row_ld4(scale + rb, x, a.W, ragged, lower);
      pu[q][0] *= su.x, pu[q][1] *= su.y, pu[q][2] *= su.z, pu[q][3] *= su.w;
      pv[q][0] *= sv.x, pv[q][1] *= sv.y, pv[q][2] *= sv.z, pv[q][3] *= sv.w;
    }
    const int e = lp(x);  // (x % 4 == 0: the four elements share a group of 16, consecutive in the padded layout)
#pragma unroll
    for (int i = 0; i < 4; ++i) bufa[e + i] = float2{pu[q][i], pv[q][i]};
  }
  __syncthreads();
  const float2* res = row_fft<-1, S, R0, R1, R2, R3>(bufa, bufb, Nx, a.f, a.tw, tid);
  float2* out = spec + (size_t)y * Nx;
  for (int x = 2 * tid; x < Nx; x += 2 * S::T) {
    const float2 c0 = res[lp(x)], c1 = res[lp(x + 1)];
    gst4(out + x, make_float4(c0.x, c0.y, c1.x, c1.y));
  }
}

struct ColsArgs {
  const float2* spec;  // [Hh][Nx]
  float2* work;        // [Ny][Nx]
  const float2* khat;  // [Nx][Ny] (column-major), normalisation folded in
  const float2* tw;
  int Hh, Nx, Ny, conj, groups;
  int keep_lo, keep_hi;  // rows [0, keep_lo) and [keep_hi, Ny) of the result are written (the others are never read)
  const FftBatch* batch;  // device memory, nullable; n_batch > 0: block b = column group b / n of dataset d0 + b % n (spec, work, khat from the table)
  int n_batch, d0;
  // Up-sampled likelihood steps (round 5): the only reader of the forward pass's result sum-pools U rows, and the adjoint
  // pass's input is U copies of every row -- both linear in the column direction too.  pool_out = U > 1: the SUMS of the
  // result's row groups [U P, U P + U) are written, to row P (groups with a kept row); pool_in = U > 1: input row r is row
  // r / U of `spec`.  Half (1 / U) of the traffic between these launches and the pooled row launch.
  int pool_out, pool_in;
  FftPasses f;
};

// columns: LANES threads per column (one wave, or two for the long columns), CB columns per block, transforms in place
// (one padded sequence per column).  R0 > 0: the radix schedule (R0, R1, R2), the length Ny = R0 R1 R2 and CB are compile
// time constants -- index arithmetic folds, and the kernel holds the five passes it runs instead of every radix in both
// directions (the generic form is 13 000 instructions, more than the instruction cache); R0 = 0: the generic form.
constexpr int ceil_div(int a, int b) { return (a + b - 1) / b; }

template <int LANES, int R0, int R1, int R2>
__device__ __forceinline__ void column_conv_static(float2* x, const float2* tw, const float2* kcol, int conj, int lane) {
  constexpr int N = R0 * R1 * R2;
  constexpr bool POW2 = (R2 & (R2 - 1)) == 0;
  column_pass_inplace<R0, -1, ceil_div(N / R0, LANES), LANES, true, lin_in(N, R0), lin_out(R0, 1)>(x, N, 1, tw, lane);
  column_pass_inplace<R1, -1, ceil_div(N / R1, LANES), LANES, true, lin_in(N, R1), lin_out(R1, R0)>(x, N, R0, tw, lane);
  column_mid_inplace<R2, ceil_div(N / R2, LANES), LANES, lin_in(N, R2)>(x, N, tw, kcol, conj, lane);
  column_pass_inplace<R1, 1, ceil_div(N / R1, LANES), LANES, POW2, lin_in(N, R1), lin_out(R1, R2)>(x, N, R2, tw, lane);
  column_pass_inplace<R0, 1, ceil_div(N / R0, LANES), LANES, POW2, lin_in(N, R0), lin_out(R0, R2 * R1)>(x, N, R2 * R1, tw, lane);
}

template <int LANES, int CBS, int R0, int R1, int R2>
__global__ __launch_bounds__((LANES * CBS > 512 ? 1024 : 512), (LANES * CBS > 512 ? 4 : 3)) void fftn_cols_kernel(ColsArgs a) {
  extern __shared__ float2 lds[];
  constexpr bool STATIC = R0 > 0;
  const int tid = threadIdx.x, lane = tid % LANES, wc = tid / LANES, CB = STATIC ? CBS : (int)blockDim.x / LANES;
  const int Ny = STATIC ? R0 * R1 * R2 : a.Ny;
  // neighbouring column groups (the same 128-byte lines of every spectrum row) go to the same XCD (blockIdx % 8), one
  // after the other: the partial lines they read and write meet in that XCD's L2
  const int nb = a.n_batch;
  const int bq = nb ? (int)blockIdx.x / nb : (int)blockIdx.x, d = nb ? a.d0 + (int)blockIdx.x - bq * nb : 0;
  const float2* const spec_in = nb ? a.batch->spec[d] : a.spec;
  float2* const work_out = nb ? a.batch->work[d] : a.work;
  const float2* const khat = nb ? a.batch->khat[d] : a.khat;
  const int per_xcd = (a.groups + 7) / 8;
  const int g = (bq % 8) * per_xcd + bq / 8;
  if (g >= a.groups) return;
  const int x0 = g * CB;
  const int stride = lp_size(Ny);
  float2* col = lds + (size_t)wc * stride;
  // ---- load: row-major pieces of CB columns, two columns (16 bytes) per thread ---------------------------------------
  const int half = CB / 2;  // float4 pieces per row
  // (CHL pieces per thread and round, loaded unconditionally and issued together, then stored to LDS: with the load under the row test the
  // compiler emitted load, full wait, LDS store per piece -- nine dependent round trips per thread for a 2304-point column)
  constexpr int CHL = 5;
  const int total = Ny * half, step = LANES * CB;
  for (int i0 = tid; i0 < total; i0 += CHL * step) {
    float4 v[CHL];
#pragma unroll
    for (int j = 0; j < CHL; ++j) {
      // (a piece past the end or in the zero padding below row Hh re-reads the thread's first piece: an L1 hit, no traffic)
      const int i = i0 + j * step < a.Hh * half ? i0 + j * step : i0;
      const int row = i / half, piece = i - row * half, rc = row < a.Hh ? row : a.Hh - 1;
      v[j] = gld4(spec_in + (size_t)(a.pool_in > 1 ? rc / a.pool_in : rc) * a.Nx + x0 + 2 * piece);
    }
#pragma unroll
    for (int j = 0; j < CHL; ++j) {
      const int i = i0 + j * step;
      if (i >= total) continue;
      const int row = i / half, piece = i - row * half;
      const float4 w = row < a.Hh ? v[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      float2* c0 = lds + (size_t)(2 * piece) * stride;
      c0[lp(row)] = float2{w.x, w.y};
      c0[stride + lp(row)] = float2{w.z, w.w};
    }
  }
  __syncthreads();
  // ---- per column: FFT, product with the kernel spectrum, inverse FFT ------------------------------------------------
  const float2* kcol = khat + (size_t)(x0 + wc) * Ny;
  if constexpr (STATIC) column_conv_static<LANES, R0, R1, R2>(col, a.tw, kcol, a.conj, lane);
  else column_conv_inplace<LANES>(col, Ny, a.f, a.tw, kcol, a.conj, lane);
  __syncthreads();
  // ---- store: rows [0, keep_lo) and [keep_hi, Ny), row-major pieces --------------------------------------------------
  if (a.pool_out > 1) {  // (uniform) sums of U rows, in row order
    const int U = a.pool_out;
    for (int i = tid; i < (Ny / U) * half; i += LANES * CB) {
      const int P = i / half, piece = i - P * half, row = U * P;
      if (row >= a.keep_lo && row + U - 1 < a.keep_hi) continue;
      const float2* c0 = lds + (size_t)(2 * piece) * stride;
      float2 p = c0[lp(row)], q = c0[stride + lp(row)];
      for (int j = 1; j < U; ++j) {
        const float2 p1 = c0[lp(row + j)], q1 = c0[stride + lp(row + j)];
        p = float2{p.x + p1.x, p.y + p1.y}, q = float2{q.x + q1.x, q.y + q1.y};
      }
      gst4(work_out + (size_t)P * a.Nx + x0 + 2 * piece, make_float4(p.x, p.y, q.x, q.y));
    }
    return;
  }
  for (int i = tid; i < Ny * half; i += LANES * CB) {
    const int row = i / half, piece = i - row * half;
    if (row >= a.keep_lo && row < a.keep_hi) continue;
    const float2* c0 = lds + (size_t)(2 * piece) * stride;
    const float2 p = c0[lp(row)], q = c0[stride + lp(row)];
    gst4(work_out + (size_t)row * a.Nx + x0 + 2 * piece, make_float4(p.x, p.y, q.x, q.y));
  }
}

// The spectrum row of a block's row pair into LDS.  The two halves of the image are convolved as separate images (real and
// imaginary part); where the convolution of one half spills across the seam into a row of the other, that row needs ONE
// part of a second inverse transform: row y (upper half) += Im c_s with s = Ny - Hh + y when y >= Hh - ra, row y + Hh (lower
// half) += Re c_s with s = Hh + y when y < rb (at most one of the two per block: Hh >= ra + rb + 1).  The part is taken
// in the Fourier domain -- for c = IFFT(C): FFT(Re c)[k] = (C[k] + conj C[-k]) / 2, FFT(Im c)[k] = (C[k] - conj C[-k]) / 2i
// -- and added to the block's own spectrum row (as a real part: + FFT(Im c_s); as an imaginary part: + i FFT(Re c_s)), so
// that a seam block runs one transform like every other block: with two, the seam blocks set the duration of the whole
// launch (every block of these launches is resident at once).
template <int T, bool ADD = false>  // ADD: buf += the row (a thread owns the same elements in every call: no barrier in between)
__device__ __forceinline__ void load_spectrum_row(float2* buf, const float2* work, int Nx, int y, int Hh, int Ny, int ra, int rb,
                                                  int tid) {
  const bool spill_up = y >= Hh - ra, spill_down = y < rb;
  const float2* src = work + (size_t)y * Nx;
  // (CH pieces per thread and round: their loads are unconditional -- a piece past the row's end re-reads the round's first --
  // and issued together, then stored to LDS.  One load, one wait, one LDS store per piece made a 4608-point row four
  // dependent round trips per thread.)
  constexpr int CH = T <= 64 ? 2 : 4;  // (one-wave rows of small fits: their launches live on blocks per CU, not on registers)
  if (!(spill_up || spill_down)) {  // (block-uniform)
    for (int x0 = 2 * tid; x0 < Nx; x0 += 2 * T * CH) {
      float4 v[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) v[j] = gld4(src + (x0 + 2 * T * j < Nx ? x0 + 2 * T * j : x0));
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int x = x0 + 2 * T * j;
        if (x >= Nx) continue;
        float2 o0 = float2{v[j].x, v[j].y}, o1 = float2{v[j].z, v[j].w};
        if (ADD) {
          const float2 p0 = buf[lp(x)], p1 = buf[lp(x + 1)];
          o0 = float2{p0.x + o0.x, p0.y + o0.y}, o1 = float2{p1.x + o1.x, p1.y + o1.y};
        }
        buf[lp(x)] = o0, buf[lp(x + 1)] = o1;
      }
    }
  } else {
    const float2* sp = work + (size_t)(spill_up ? Ny - Hh + y : Hh + y) * Nx;
    for (int x0 = 2 * tid; x0 < Nx; x0 += 2 * T * CH) {
      float4 vv[CH], cc[CH];
      float2 mm0[CH], mm1[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int x = x0 + 2 * T * j < Nx ? x0 + 2 * T * j : x0;
        vv[j] = gld4(src + x), cc[j] = gld4(sp + x);
        mm0[j] = gld2(sp + (x == 0 ? 0 : Nx - x)), mm1[j] = gld2(sp + (Nx - x - 1));  // C[-x], C[-(x + 1)]
      }
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int x = x0 + 2 * T * j;
        if (x >= Nx) continue;
        const float4 v = vv[j], c = cc[j];
        const float2 m0 = mm0[j], m1 = mm1[j];
        float2 o0, o1;
        if (spill_up) {  // + (C[k] - conj C[-k]) / 2i = ((Im C[k] + Im C[-k]) / 2, -(Re C[k] - Re C[-k]) / 2)
          o0 = float2{v.x + 0.5f * (c.y + m0.y), v.y - 0.5f * (c.x - m0.x)};
          o1 = float2{v.z + 0.5f * (c.w + m1.y), v.w - 0.5f * (c.z - m1.x)};
        } else {  // + i (C[k] + conj C[-k]) / 2 = (-(Im C[k] - Im C[-k]) / 2, (Re C[k] + Re C[-k]) / 2)
          o0 = float2{v.x - 0.5f * (c.y - m0.y), v.y + 0.5f * (c.x + m0.x)};
          o1 = float2{v.z - 0.5f * (c.w - m1.y), v.w + 0.5f * (c.z + m1.x)};
        }
        if (ADD) {
          const float2 p0 = buf[lp(x)], p1 = buf[lp(x + 1)];
          o0 = float2{p0.x + o0.x, p0.y + o0.y}, o1 = float2{p1.x + o1.x, p1.y + o1.y};
        }
        buf[lp(x)] = o0, buf[lp(x + 1)] = o1;
      }
    }
  }
  __syncthreads();
}

// The same in two steps, for a block that runs several transforms in a row: the (combined) spectrum row into REGISTERS --
// the loads of the next row are in flight while the block transforms the current one -- and from there into LDS.
template <int T, int ROW_PRE>  // ROW_PRE: float4 pieces (two spectrum elements) of a row per thread
__device__ __forceinline__ void load_spectrum_row_regs(float4 (&pre)[ROW_PRE], const float2* work, int Nx, int y, int Hh, int Ny, int ra,
                                                       int rb, int tid) {
  const bool spill_up = y >= Hh - ra, spill_down = y < rb;
  const float2* src = work + (size_t)y * Nx;
  const float2* sp = work + (size_t)(spill_up ? Ny - Hh + y : Hh + y) * Nx;
#pragma unroll
  for (int i = 0; i < ROW_PRE; ++i) {
    const int x = 2 * (tid + i * T);
    if (x >= Nx) continue;
    float4 v = gld4(src + x);
    if (spill_up || spill_down) {  // (block-uniform; see load_spectrum_row)
      const float4 c = gld4(sp + x);
      const float2 m0 = gld2(sp + (x == 0 ? 0 : Nx - x)), m1 = gld2(sp + (Nx - x - 1));
      if (spill_up) {
        v = make_float4(v.x + 0.5f * (c.y + m0.y), v.y - 0.5f * (c.x - m0.x), v.z + 0.5f * (c.w + m1.y), v.w - 0.5f * (c.z - m1.x));
      } else {
        v = make_float4(v.x - 0.5f * (c.y - m0.y), v.y + 0.5f * (c.x + m0.x), v.z - 0.5f * (c.w - m1.y), v.w + 0.5f * (c.z + m1.x));
      }
    }
    pre[i] = v;
  }
}

template <int T, int ROW_PRE>
__device__ __forceinline__ void store_spectrum_row_regs(float2* buf, const float4 (&pre)[ROW_PRE], int Nx, int tid) {
#pragma unroll
  for (int i = 0; i < ROW_PRE; ++i) {
    const int x = 2 * (tid + i * T);
    if (x >= Nx) continue;
    buf[lp(x)] = float2{pre[i].x, pre[i].y}, buf[lp(x + 1)] = float2{pre[i].z, pre[i].w};
  }
  __syncthreads();
}

struct RowsInvArgs {
  const float2* work;  // [Ny][Nx]
  const float2* tw;
  float* out;          // (H, W)
  const float* scale;  // adjoint: nullable
  int H, W, Hh, Nx, Ny, ra, rb;  // the convolution of a half spills ra rows above and rb rows below it
  float coef;
  int accumulate;
  // the loss of a likelihood step: block 0 sums the fin_count partial sums the middle launch left (the arithmetic and
  // order of finalize_sum_kernel) -- one dependent launch less per dataset
  const double* fin_partials;
  int fin_count;
  double fin_scale, fin_offset;
  float* fin_out;
  const double* fin2_partials;  // a second sum of fin_count terms (block 1): d loss / d log background norm
  double fin2_scale;
  float* fin2_out;
  const FftBatch* batch;        // fftn_rows_inv_batch_kernel (device memory): the n_batch datasets whose adjoints one block adds up, in order
  int n_batch, d0;              // (fftn_rows_inv_kernel: block b = row pair b / n of dataset d0 + b % n)
  FftPasses f;
};

// rows^-1 + epilogue.  ADJ = false: out = conv;  ADJ = true: out (+)= (coef * corr) * scale
template <bool ADJ, int R0, int R1, int R2, int R3>
__global__ __launch_bounds__((RowSched<R0, R1, R2, R3>::T), (RowSched<R0, R1, R2, R3>::WAVES_PER_SIMD)) void fftn_rows_inv_kernel(RowsInvArgs a) {
#pragma clang fp contract(off)  // (product, product, sum: the batched form below must give the same bits)
  extern __shared__ float2 lds[];
  using S = RowSched<R0, R1, R2, R3>;
  const int Nx = S::STATIC ? S::N : a.Nx;
  // n_batch > 0 (ADJ, calibrated batched step): block b = row pair b / n of dataset b % n; out = the dataset's gshift image
  // (overwritten), scale = its exposure, and the blocks of row pairs 0 / 1 finalise the dataset's loss / norm gradient
  const int tid = threadIdx.x, nb = ADJ ? a.n_batch : 0;
  const int y = nb ? (int)blockIdx.x / (ADJ ? nb : 1) : (int)blockIdx.x, d = nb ? a.d0 + (int)blockIdx.x - y * nb : 0;
  const float2* const work = nb ? a.batch->work[d] : a.work;
  const float* const scale = nb ? a.batch->exposure[d] : a.scale;
  float* const out = nb ? a.batch->gshift[d] : a.out;
  float2* bufa = lds;
  float2* bufb = lds + lp_size(Nx);
  constexpr int MAXQ = S::MAXQ;
  const size_t o1 = (size_t)y * a.W, o2 = (size_t)(y + a.Hh) * a.W;
  const bool ragged = (a.W & 3) != 0, lower = y + a.Hh < a.H;
  // the epilogue's exposure rows do not depend on the transform: requested FIRST, so that they arrive while the block loads
  // its spectrum row and transforms it, not in a round trip of their own behind it
  // (not the one-wave rows of small fits: four pieces per thread there, and their launches live on blocks per CU)
  constexpr bool PRE = ADJ && S::T > 64;
  float4 s1[PRE ? MAXQ : 1], s2[PRE ? MAXQ : 1];
  if (PRE) {
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
      const int x = 4 * (tid + q * S::T), xs = x < a.W ? x : 0;
      s1[q] = s2[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (scale) s1[q] = row_ld4(scale + o1, xs, a.W, ragged), s2[q] = row_ld4(scale + o2, xs, a.W, ragged, lower);
    }
  }
  load_spectrum_row<S::T>(bufa, work, Nx, y, a.Hh, a.Ny, a.ra, a.rb, tid);
  const float2* r = row_fft<1, S, R0, R1, R2, R3>(bufa, bufb, Nx, a.f, a.tw, tid);
#pragma unroll
  for (int q = 0; q < MAXQ; ++q) {
    const int x = 4 * (tid + q * S::T);
    if (x >= a.W) continue;
    const int e = lp(x);
    const float2 c0 = r[e], c1 = r[e + 1], c2 = r[e + 2], c3 = r[e + 3];
    float4 up = make_float4(c0.x, c1.x, c2.x, c3.x), dn = make_float4(c0.y, c1.y, c2.y, c3.y);
    if (ADJ) {
      up.x *= a.coef, up.y *= a.coef, up.z *= a.coef, up.w *= a.coef;
      dn.x *= a.coef, dn.y *= a.coef, dn.z *= a.coef, dn.w *= a.coef;
      if (scale) {
        const float4 e1 = PRE ? s1[PRE ? q : 0] : row_ld4(scale + o1, x, a.W, ragged);
        const float4 e2 = PRE ? s2[PRE ? q : 0] : row_ld4(scale + o2, x, a.W, ragged, lower);
        up.x *= e1.x, up.y *= e1.y, up.z *= e1.z, up.w *= e1.w;
        dn.x *= e2.x, dn.y *= e2.y, dn.z *= e2.z, dn.w *= e2.w;
      }
      if (a.accumulate && !nb) {
        const float4 p1 = row_ld4(out + o1, x, a.W, ragged), p2 = row_ld4(out + o2, x, a.W, ragged, lower);
        up.x += p1.x, up.y += p1.y, up.z += p1.z, up.w += p1.w;
        dn.x += p2.x, dn.y += p2.y, dn.z += p2.z, dn.w += p2.w;
      }
    }
    row_st4(out + o1, x, a.W, ragged, up);
    row_st4(out + o2, x, a.W, ragged, dn, lower);
  }
  if (ADJ && a.fin_partials && y == 0) {  // (block-uniform)
    __shared__ double red[S::T / 64];
    const double* part = a.fin_partials + (size_t)d * a.fin_count;
    double acc = 0.0;
    for (int i = tid; i < a.fin_count; i += S::T) acc += part[i];
    const double total = block_sum<S::T>(acc, red);
    if (tid == 0) {
      if (nb) a.batch->loss_out[d][0] = (float)(a.fin_scale * total + (double)a.batch->loss_offset[d]);
      else a.fin_out[0] = (float)(a.fin_scale * total + a.fin_offset);
    }
  }
  if (ADJ && a.fin2_partials && y == 1 && (!nb || a.batch->grad_log_bkg_norm[d])) {
    __shared__ double red2[S::T / 64];
    const double* part = a.fin2_partials + (size_t)d * a.fin_count;
    double acc = 0.0;
    for (int i = tid; i < a.fin_count; i += S::T) acc += part[i];
    const double total = block_sum<S::T>(acc, red2);
    if (tid == 0) (nb ? a.batch->grad_log_bkg_norm[d] : a.fin2_out)[0] = (float)(a.fin2_scale * total);
  }
}

// The adjoint's last launch for SEVERAL datasets: block y runs the inverse row transform of every dataset's row pair in
// turn and adds (coef * corr_d) * exposure_d in dataset order -- the sums the per-dataset launches leave in `out` when
// each accumulates onto its predecessor, bit for bit -- and writes the two gradient rows once.  Blocks 0 .. n - 1
// finalise the datasets' losses (fin_count partial sums each).
template <int R0, int R1, int R2, int R3>
__global__ __launch_bounds__((RowSched<R0, R1, R2, R3>::T), (RowSched<R0, R1, R2, R3>::WAVES_PER_SIMD)) void fftn_rows_inv_batch_kernel(RowsInvArgs a) {
#pragma clang fp contract(off)
  extern __shared__ float2 lds[];
  using S = RowSched<R0, R1, R2, R3>;
  const int Nx = S::STATIC ? S::N : a.Nx;
  const int tid = threadIdx.x, y = blockIdx.x;
  float2* bufa = lds;
  float2* bufb = lds + lp_size(Nx);
  constexpr int MAXQ = S::MAXQ;
  const size_t o1 = (size_t)y * a.W, o2 = (size_t)(y + a.Hh) * a.W;
  const bool ragged = (a.W & 3) != 0, lower = y + a.Hh < a.H;
  float4 au[MAXQ], ad[MAXQ];
#pragma unroll
  for (int q = 0; q < MAXQ; ++q) au[q] = ad[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 pre[S::PRE];
  load_spectrum_row_regs<S::T, S::PRE>(pre, a.batch->work[0], Nx, y, a.Hh, a.Ny, a.ra, a.rb, tid);
#pragma unroll 1
  for (int d = 0; d < a.n_batch; ++d) {
    store_spectrum_row_regs<S::T, S::PRE>(bufa, pre, Nx, tid);
    if (d + 1 < a.n_batch) load_spectrum_row_regs<S::T, S::PRE>(pre, a.batch->work[d + 1], Nx, y, a.Hh, a.Ny, a.ra, a.rb, tid);
    const float2* r = row_fft<1, S, R0, R1, R2, R3>(bufa, bufb, Nx, a.f, a.tw, tid);
    const float* scale = a.batch->exposure[d];
    const bool add = d > 0 || a.accumulate;
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
      const int x = 4 * (tid + q * S::T);
      if (x >= a.W) continue;
      const int e = lp(x);
      const float2 c0 = r[e], c1 = r[e + 1], c2 = r[e + 2], c3 = r[e + 3];
      float4 up = make_float4(c0.x, c1.x, c2.x, c3.x), dn = make_float4(c0.y, c1.y, c2.y, c3.y);
      up.x *= a.coef, up.y *= a.coef, up.z *= a.coef, up.w *= a.coef;
      dn.x *= a.coef, dn.y *= a.coef, dn.z *= a.coef, dn.w *= a.coef;
      const float4 s1 = row_ld4(scale + o1, x, a.W, ragged), s2 = row_ld4(scale + o2, x, a.W, ragged, lower);
      up.x *= s1.x, up.y *= s1.y, up.z *= s1.z, up.w *= s1.w;
      dn.x *= s2.x, dn.y *= s2.y, dn.z *= s2.z, dn.w *= s2.w;
      if (add) {
        float4 g1 = au[q], g2 = ad[q];
        if (d == 0) g1 = row_ld4(a.out + o1, x, a.W, ragged), g2 = row_ld4(a.out + o2, x, a.W, ragged, lower);
        up.x += g1.x, up.y += g1.y, up.z += g1.z, up.w += g1.w;
        dn.x += g2.x, dn.y += g2.y, dn.z += g2.z, dn.w += g2.w;
      }
      au[q] = up, ad[q] = dn;
    }
    __syncthreads();  // the result buffer is the next dataset's work space
  }
#pragma unroll
  for (int q = 0; q < MAXQ; ++q) {
    const int x = 4 * (tid + q * S::T);
    if (x >= a.W) continue;
    row_st4(a.out + o1, x, a.W, ragged, au[q]);
    row_st4(a.out + o2, x, a.W, ragged, ad[q], lower);
  }
  if (a.fin_partials) {
    __shared__ double red[S::T / 64];
    for (int d = blockIdx.x; d < a.n_batch; d += gridDim.x) {  // (block-uniform; more datasets than blocks: several per block)
      const double* part = a.fin_partials + (size_t)d * a.fin_count;
      double acc = 0.0;
      for (int i = tid; i < a.fin_count; i += S::T) acc += part[i];
      const double total = block_sum<S::T>(acc, red);
      if (tid == 0) a.batch->loss_out[d][0] = (float)(a.fin_scale * total + (double)a.batch->loss_offset[d]);
      __syncthreads();
    }
  }
}

struct RowsPoissonArgs {
  const float2* work;  // [Ny][Nx]: the forward convolution after the column pass
  float2* spec;        // [Hh][Nx]: <- row spectra of g (the input of the adjoint's column pass)
  const float2* tw;
  const float* background;
  const float* counts;
  double* partials;    // [Hh] ([n][Hh] for a batch)
  int H, W, Hh, Nx, Ny, ra, rb;
  float eps, inv_n;
  const FftBatch* batch;  // device memory, nullable; n_batch > 0: block b = row pair b / n of dataset d0 + b % n
  int n_batch, d0;
  FftPasses f;
};

// The middle of a likelihood step in ONE launch: rows^-1 of the forward convolution, the Poisson pass on the two
// finished image rows (clip, + background, NLL term, g = d loss / d conv where conv >= 0: `poisson_point`, the
// arithmetic of every Poisson pass of the library), and at once the forward row transform of the g rows -- they ARE the
// row pair the adjoint's first launch would read.  The convolution image and the g image never exist.
template <int R0, int R1, int R2, int R3>
__global__ __launch_bounds__((RowSched<R0, R1, R2, R3>::T), (RowSched<R0, R1, R2, R3>::WAVES_PER_SIMD)) void fftn_rows_poisson_kernel(RowsPoissonArgs a) {
  extern __shared__ float2 lds[];
  using S = RowSched<R0, R1, R2, R3>;
  __shared__ double red[S::T / 64];
  const int Nx = S::STATIC ? S::N : a.Nx;
  const int tid = threadIdx.x, nb = a.n_batch;
  const int y = nb ? (int)blockIdx.x / nb : (int)blockIdx.x, d = nb ? a.d0 + (int)blockIdx.x - y * nb : 0;
  const float2* const work = nb ? a.batch->work[d] : a.work;
  float2* const spec = nb ? a.batch->spec[d] : a.spec;
  const float* const background = nb ? a.batch->background[d] : a.background;
  const float* const counts = nb ? a.batch->counts[d] : a.counts;
  float2* bufa = lds;
  float2* bufb = lds + lp_size(Nx);
  constexpr int MAXQ = S::MAXQ;
  load_spectrum_row<S::T>(bufa, work, Nx, y, a.Hh, a.Ny, a.ra, a.rb, tid);
  const float2* r = row_fft<1, S, R0, R1, R2, R3>(bufa, bufb, Nx, a.f, a.tw, tid);
  const size_t o1 = (size_t)y * a.W, o2 = (size_t)(y + a.Hh) * a.W;
  const bool ragged = (a.W & 3) != 0, lower = y + a.Hh < a.H;
  float4 gu[MAXQ], gd[MAXQ];
  float local = 0.f;
#pragma unroll
  for (int q = 0; q < MAXQ; ++q) {
    const int x = 4 * (tid + q * S::T);
    gu[q] = gd[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (x >= a.W) continue;
    const int e = lp(x);
    const float2 c0 = r[e], c1 = r[e + 1], c2 = r[e + 2], c3 = r[e + 3];
    float up[4] = {c0.x, c1.x, c2.x, c3.x}, dn[4] = {c0.y, c1.y, c2.y, c3.y};
    const float4 b1 = row_ld4(background + o1, x, a.W, ragged), b2 = row_ld4(background + o2, x, a.W, ragged, lower);
    const float4 n1 = row_ld4(counts + o1, x, a.W, ragged), n2 = row_ld4(counts + o2, x, a.W, ragged, lower);
    const float bu[4] = {b1.x, b1.y, b1.z, b1.w}, bd[4] = {b2.x, b2.y, b2.z, b2.w};
    const float cu[4] = {n1.x, n1.y, n1.z, n1.w}, cd[4] = {n2.x, n2.y, n2.z, n2.w};
    float g1[4], g2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float term, g;
      const bool inside = !ragged || x + i < a.W;  // (pixels of the padding carry no term and no gradient)
      poisson_point(fmaxf(up[i], 0.f) + bu[i], cu[i], a.eps, a.inv_n, term, g);
      local += inside ? term : 0.f;
      g1[i] = inside && up[i] >= 0.f ? g : 0.f;  // clamp backward
      poisson_point(fmaxf(dn[i], 0.f) + bd[i], cd[i], a.eps, a.inv_n, term, g);
      local += inside && lower ? term : 0.f;
      g2[i] = inside && lower && dn[i] >= 0.f ? g : 0.f;
    }
    gu[q] = make_float4(g1[0], g1[1], g1[2], g1[3]), gd[q] = make_float4(g2[0], g2[1], g2[2], g2[3]);
  }
  const double total = block_sum<S::T>((double)local, red);
  if (tid == 0) a.partials[(size_t)d * a.Hh + y] = total;
  __syncthreads();  // every thread has read its part of the convolution row: the buffers are free
  // ---- z = g[y] + i g[y + Hh], zero padded: the adjoint's row transform ---------------------------------------------
#pragma unroll
  for (int q = 0; q < MAXQ; ++q) {  // (Nx <= 4608 < 4 * 256 * MAXQ)
    const int x = 4 * (tid + q * S::T), e = lp(x);
    if (x >= Nx) continue;
    float4 u = make_float4(0.f, 0.f, 0.f, 0.f), v = u;
    if (x < a.W) u = gu[q], v = gd[q];
    bufa[e] = float2{u.x, v.x}, bufa[e + 1] = float2{u.y, v.y}, bufa[e + 2] = float2{u.z, v.z}, bufa[e + 3] = float2{u.w, v.w};
  }
  __syncthreads();
  const float2* res = row_fft<-1, S, R0, R1, R2, R3>(bufa, bufb, Nx, a.f, a.tw, tid);
  float2* out = spec + (size_t)y * Nx;
  for (int x = 2 * tid; x < Nx; x += 2 * S::T) {
    const float2 c0 = res[lp(x)], c1 = res[lp(x + 1)];
    gst4(out + x, make_float4(c0.x, c0.y, c1.x, c1.y));
  }
}

struct RowsPooledArgs {
  const float2* work;  // [Ny][Nx]: the forward convolution on the flux grid after the column pass
  float2* spec;        // [Hh][Nx]: <- row spectra of the up-sampled g
  const float2* tw;
  const float* background;    // counts grid (H / U, W / U)
  const float* counts;
  const float* log_bkg_norm;  // nullable, device [1]: background *= exp(.) (models/npred.py:236-239)
  double* partials;           // [Hh / U]: block sums of the Poisson terms
  double* partials_b;         // nullable [Hh / U]: block sums of g * background (d loss / d log norm up to the scale)
  int H, W, Hh, Nx, Ny, ra, rb;
  int pooled_io;              // 1: `work` holds the row-group sums (ColsArgs::pool_out) and `spec` takes ONE row per counts row
  float eps, inv_n;
  const FftBatch* batch;      // device memory, nullable; n_batch > 0: block b = counts-row pair b / n of dataset d0 + b % n
  int n_batch, d0;
  FftPasses f;
};

// The middle of a likelihood step with up-sampling (models/npred.py:181-191: sum-pool the convolution over U x U flux
// pixels, THEN clip) in one launch: block Y owns the counts rows Y and Y + H / (2 U): U inverse row transforms (flux rows
// U Y + j of both halves), pooled sums in registers, the Poisson pass on the two counts rows, and ONE forward row
// transform of the up-sampled g rows -- the U flux rows of a counts row carry the same g, so their spectrum row is
// computed once and stored U times.  Replaces rows^-1 -> convolution image -> pooled Poisson kernel -> g image -> rows.
template <int U, int R0, int R1, int R2, int R3>
__global__ __launch_bounds__((RowSched<R0, R1, R2, R3>::T), (RowSched<R0, R1, R2, R3>::T == 576 ? 5 : RowSched<R0, R1, R2, R3>::STATIC ? 4 : 1)) void fftn_rows_pooled_kernel(RowsPooledArgs a) {
  extern __shared__ float2 lds[];
  using S = RowSched<R0, R1, R2, R3>;
  __shared__ double red[S::T / 64];
  const int Nx = S::STATIC ? S::N : a.Nx;
  const int tid = threadIdx.x, nb = a.n_batch;
  const int Y = nb ? (int)blockIdx.x / nb : (int)blockIdx.x, d = nb ? a.d0 + (int)blockIdx.x - Y * nb : 0;
  const float2* const work = nb ? a.batch->work[d] : a.work;
  float2* const spec = nb ? a.batch->spec[d] : a.spec;
  const float* const background = nb ? a.batch->background[d] : a.background;
  const float* const counts = nb ? a.batch->counts[d] : a.counts;
  const float* const log_bkg_norm = nb ? a.batch->log_bkg_norm[d] : a.log_bkg_norm;
  const size_t pbase = (size_t)d * (a.Hh / U);
  float2* bufa = lds;
  float2* bufb = lds + lp_size(Nx);
  // A thread owns PIECES of PX = lcm(4, U) flux pixels = G counts pixels of the block's two counts rows (round 5: any
  // up-sampling factor; U = 2: 4 flux pixels / 2 counts pixels, U = 4: 4 / 1, U = 3: 12 / 4).  The flux pixels of a piece
  // are read from the transform's result in groups of four (one group of 16 in the padded layout: contiguous).
  constexpr int PX = U % 4 == 0 ? U : U % 2 == 0 ? 2 * U : 4 * U, G = PX / U, SUB = PX / 4;
  constexpr int MAXP = (S::LIMIT + PX * S::T - 1) / (PX * S::T);
  float pu[MAXP][G], pd[MAXP][G];  // pooled sums: counts row Y (upper half) and Y + H / (2 U) (lower half)
#pragma unroll
  for (int q = 0; q < MAXP; ++q)
#pragma unroll
    for (int c = 0; c < G; ++c) pu[q][c] = pd[q][c] = 0.f;
  const int Wd = a.W / U, Hdh = a.Hh / U;
  // The sum-pool over the U flux rows of a counts row is linear and so is the row transform: the U spectrum rows are ADDED
  // in the Fourier domain (each with its seam term) and ONE inverse transform gives the row sums -- until the middle of
  // round 5 a block ran U inverse transforms and added their results (c6: three dependent 4608-point transforms per block,
  // now two).  The pool over x is summed left to right from the transform's result.
  if (a.pooled_io) {  // (uniform) the column pass has summed the rows: row Y of the pooled array, its seam rows in pooled units
    load_spectrum_row<S::T>(bufa, work, Nx, Y, a.Hh / U, a.Ny / U, (a.ra + U - 1) / U, (a.rb + U - 1) / U, tid);
  } else {
    load_spectrum_row<S::T>(bufa, work, Nx, U * Y, a.Hh, a.Ny, a.ra, a.rb, tid);
#pragma unroll 1
    for (int j = 1; j < U; ++j) load_spectrum_row<S::T, true>(bufa, work, Nx, U * Y + j, a.Hh, a.Ny, a.ra, a.rb, tid);
  }
  // background and counts of the thread's counts pixels: unconditional global loads (clamped to the row), issued here so that
  // they are in flight while the pooled sums are read from LDS (under the bounds tests, through the table's generic
  // pointers, they were flat loads with a full wait each, behind the transform)
  float bgu[MAXP][G], bgd[MAXP][G], cnu[MAXP][G], cnd[MAXP][G];
  {
    const float2* r = row_fft<1, S, R0, R1, R2, R3>(bufa, bufb, Nx, a.f, a.tw, tid);
#pragma unroll
    for (int q = 0; q < MAXP; ++q) {
      const int x = PX * (tid + q * S::T);
#pragma unroll
      for (int c = 0; c < G; ++c) {
        const int xc = min((x < a.W ? x : 0) / U + c, Wd - 1);
        const size_t o1 = (size_t)Y * Wd + xc, o2 = (size_t)(Y + Hdh) * Wd + xc;
        bgu[q][c] = gld(background + o1), bgd[q][c] = gld(background + o2);
        cnu[q][c] = gld(counts + o1), cnd[q][c] = gld(counts + o2);
      }
    }
#pragma unroll
    for (int q = 0; q < MAXP; ++q) {
      const int x = PX * (tid + q * S::T);
      if (x >= a.W) continue;
#pragma unroll
      for (int sub = 0; sub < SUB; ++sub) {
        const int e = lp(x + 4 * sub);
        const float2 v[4] = {r[e], r[e + 1], r[e + 2], r[e + 3]};
#pragma unroll
        for (int i = 0; i < 4; ++i) pu[q][(4 * sub + i) / U] += v[i].x, pd[q][(4 * sub + i) / U] += v[i].y;
      }
    }
    __syncthreads();  // the result buffer is the next transform's work space
  }
  const float norm = log_bkg_norm ? expf(gld(log_bkg_norm)) : 1.f;
  float gu[MAXP][G], gd[MAXP][G];
  double local = 0.0, local_b = 0.0;
#pragma unroll
  for (int q = 0; q < MAXP; ++q) {
    const int x = PX * (tid + q * S::T);
#pragma unroll
    for (int c = 0; c < G; ++c) gu[q][c] = gd[q][c] = 0.f;
    if (x >= a.W) continue;
#pragma unroll
    for (int c = 0; c < G; ++c) {
      if (x / U + c >= Wd) continue;  // (the last piece of a row may hold fewer counts pixels)
      const float b1 = log_bkg_norm ? bgu[q][c] * norm : bgu[q][c];
      const float b2 = log_bkg_norm ? bgd[q][c] * norm : bgd[q][c];
      float term, g;
      poisson_point(fmaxf(pu[q][c], 0.f) + b1, cnu[q][c], a.eps, a.inv_n, term, g);
      local += (double)term, local_b += (double)(g * b1);
      gu[q][c] = pu[q][c] >= 0.f ? g : 0.f;  // clamp backward
      poisson_point(fmaxf(pd[q][c], 0.f) + b2, cnd[q][c], a.eps, a.inv_n, term, g);
      local += (double)term, local_b += (double)(g * b2);
      gd[q][c] = pd[q][c] >= 0.f ? g : 0.f;
    }
  }
  const double total = block_sum<S::T>(local, red);
  if (tid == 0) a.partials[pbase + Y] = total;
  if (a.partials_b) {
    __syncthreads();
    const double total_b = block_sum<S::T>(local_b, red);
    if (tid == 0) a.partials_b[pbase + Y] = total_b;
  }
  // ---- z = g_up[y] + i g_up[y + Hh], zero padded: the adjoint's row transform, the same for the U flux rows ---------------
#pragma unroll
  for (int q = 0; q < MAXP; ++q) {
    const int x = PX * (tid + q * S::T);
#pragma unroll
    for (int sub = 0; sub < SUB; ++sub) {
      if (x + 4 * sub >= Nx) continue;
      const int e = lp(x + 4 * sub);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        bufa[e + i] = x + 4 * sub + i < a.W ? float2{gu[q][(4 * sub + i) / U], gd[q][(4 * sub + i) / U]} : float2{0.f, 0.f};
    }
  }
  __syncthreads();
  const float2* res = row_fft<-1, S, R0, R1, R2, R3>(bufa, bufb, Nx, a.f, a.tw, tid);
  for (int x = 2 * tid; x < Nx; x += 2 * S::T) {
    const float2 c0 = res[lp(x)], c1 = res[lp(x + 1)];
    const float4 v = make_float4(c0.x, c0.y, c1.x, c1.y);
    if (a.pooled_io) {
      gst4(spec + (size_t)Y * Nx + x, v);  // (the adjoint's column pass reads it for its U rows)
    } else {
#pragma unroll
      for (int j = 0; j < U; ++j) gst4(spec + (size_t)(U * Y + j) * Nx + x, v);
    }
  }
}

// Kernel spectrum, column-major and normalised: khat[x][v] = 1 / (Ny Nx) sum_j sum_i psf[j][i] exp(-2 pi i (v (j - oy) / Ny +
// x (i - ox) / Nx)) -- the PSF placed on the (Ny, Nx) torus with its centre tap at the origin, so that the circular
// convolution IS the 'same' crop of utils/torch.py:337-344.  Two small DFT stages in double precision (set-up time only).
__global__ __launch_bounds__(256) void fftn_spectrum_rows_kernel(const float* psf, double2* tmp, int kh, int kw, int ox, int Nx) {
  const int x = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
  if (x >= Nx) return;
  double re = 0.0, im = 0.0;
  for (int i = 0; i < kw; ++i) {
    const long m = ((long)x * (i - ox)) % Nx;
    double s, c;
    sincospi(-2.0 * (double)m / (double)Nx, &s, &c);
    const double p = (double)psf[j * kw + i];
    re += p * c, im += p * s;
  }
  tmp[(size_t)j * Nx + x] = double2{re, im};
}

__global__ __launch_bounds__(256) void fftn_spectrum_cols_kernel(const double2* tmp, float2* khat, int kh, int oy, int Nx, int Ny) {
  const int v = blockIdx.x * 256 + threadIdx.x, x = blockIdx.y;
  if (v >= Ny) return;
  double re = 0.0, im = 0.0;
  for (int j = 0; j < kh; ++j) {
    const long m = ((long)v * (j - oy)) % Ny;
    double s, c;
    sincospi(-2.0 * (double)m / (double)Ny, &s, &c);
    const double2 t = tmp[(size_t)j * Nx + x];
    re += t.x * c - t.y * s, im += t.x * s + t.y * c;
  }
  const double norm = 1.0 / ((double)Nx * (double)Ny);
  khat[(size_t)x * Ny + v] = float2{(float)(re * norm), (float)(im * norm)};
}

FftPasses passes_of(int N) {
  const Radices r = factorize(N);
  FftPasses f{};
  f.n = r.n;
  for (int i = 0; i < r.n; ++i) f.r[i] = r.r[i];
  return f;
}

}  // namespace

bool fftn_supported(int H, int W, int kh, int kw) {
  // any image size (round 5): an odd number of rows is a lower half that is one row short, a width that is not a
  // multiple of 4 is read and written at 4-byte alignment (row_ld4 / row_st4)
  const int Hh = (H + 1) / 2;
  if (H < 2 || Hh < kh || kh < 1 || kw < 1) return false;
  if (W < 8) return false;  // (the five-pixel windows of a calibrated row load, issue_row5, need W >= 5; narrower images: rocFFT)
  if (W > 4 * ROW_THREADS * 5) return false;
  const int ox = (kw - 1) / 2;
  const int nx = next_length(W + std::max(ox, kw - 1 - ox)), ny = next_length(Hh + kh - 1, false);
  // LDS: two padded sequences per row block, one per wave of a column block
  if (!(nx > 0 && ny > 0 && nx <= 4608 && ny <= 2304 && ny >= 2 * kh)) return false;
  return factorize(nx).n > 0 && column_lanes(ny, passes_of(ny)) != 0;
}

int fftn_create(FftNative* n, int H, int W, int kh, int kw) {
  *n = FftNative{};
  n->H = H, n->W = W, n->kh = kh, n->kw = kw, n->oy = (kh - 1) / 2, n->ox = (kw - 1) / 2;
  n->Hh = (H + 1) / 2;  // (odd H: the lower half is one row short)
  n->Nx = next_length(W + std::max(n->ox, kw - 1 - n->ox));
  n->Ny = next_length(n->Hh + kh - 1, false);
  JD_HIP(hipMalloc(&n->spec, (size_t)n->Hh * n->Nx * sizeof(float2)));
  JD_HIP(hipMalloc(&n->work, (size_t)n->Ny * n->Nx * sizeof(float2)));
  std::vector<float2> tw;
  for (int pass = 0; pass < 2; ++pass) {
    const int N = pass ? n->Ny : n->Nx;
    tw.resize(N);
    for (int m = 0; m < N; ++m) {
      const double ang = -2.0 * M_PI * (double)m / (double)N;
      tw[m] = float2{(float)std::cos(ang), (float)std::sin(ang)};
    }
    float2** dst = pass ? &n->tw_y : &n->tw_x;
    JD_HIP(hipMalloc(dst, (size_t)N * sizeof(float2)));
    JD_HIP(hipMemcpy(*dst, tw.data(), (size_t)N * sizeof(float2), hipMemcpyHostToDevice));
  }
  return JD_OK;
}

void fftn_destroy(FftNative* n) {
  if (n->spec) (void)hipFree(n->spec);
  if (n->work) (void)hipFree(n->work);
  if (n->tw_x) (void)hipFree(n->tw_x);
  if (n->tw_y) (void)hipFree(n->tw_y);
  *n = FftNative{};
}

size_t fftn_spectrum_elements(const FftNative& n) { return (size_t)n.Nx * n.Ny; }

int fftn_spectrum(const FftNative& n, const float* psf, float2* khat, hipStream_t stream) {
  // (the double-precision row stage borrows the column work buffer: kh * Nx * 16 bytes <= Ny * Nx * 8)
  double2* tmp = reinterpret_cast<double2*>(n.work);
  fftn_spectrum_rows_kernel<<<dim3((n.Nx + 255) / 256, n.kh), 256, 0, stream>>>(psf, tmp, n.kh, n.kw, n.ox, n.Nx);
  JD_LAUNCH_CHECK();
  fftn_spectrum_cols_kernel<<<dim3((n.Ny + 255) / 256, n.Nx), 256, 0, stream>>>(tmp, khat, n.kh, n.oy, n.Nx, n.Ny);
  JD_LAUNCH_CHECK();
  return JD_OK;
}

namespace {
// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (device, kernel): remembered per pair, under a lock (a
// process may drive several devices, from several host threads)
int lds_attr(const void* kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return JD_OK;
  static std::mutex mutex;
  static std::map<std::pair<int, const void*>, size_t> set;
  int device = 0;
  JD_HIP(hipGetDevice(&device));
  std::lock_guard<std::mutex> lock(mutex);
  size_t& have = set[{device, kernel}];
  if (bytes > have) {
    JD_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    have = bytes;
  }
  return JD_OK;
}

int launch_cols(const FftNative& n, const float2* khat, int adjoint, hipStream_t stream, const FftBatch* batch = nullptr, int n_batch = 0,
                int d0 = 0, int pool = 1) {
  const FftPasses fy = passes_of(n.Ny);
  const int ra = adjoint ? n.kh - 1 - n.oy : n.oy, rb = adjoint ? n.oy : n.kh - 1 - n.oy;
  ColsArgs a{};
  a.spec = n.spec, a.work = n.work, a.khat = khat, a.tw = n.tw_y, a.Hh = n.Hh, a.Nx = n.Nx, a.Ny = n.Ny, a.conj = adjoint ? 1 : 0;
  a.keep_lo = n.Hh + rb, a.keep_hi = n.Ny - ra, a.f = fy;
  a.batch = batch, a.n_batch = n_batch, a.d0 = d0;
  a.pool_out = adjoint ? 1 : pool, a.pool_in = adjoint ? pool : 1;
  const size_t per_col = (size_t)lp_size(n.Ny) * sizeof(float2);
  const int lanes = column_lanes(n.Ny, fy);
  if (!lanes) return fail(JD_ERR_INVALID, "native FFT: no column kernel for length %d", n.Ny);
  // columns per block: 4 (32 contiguous bytes of every spectrum row).  Measured at 2048^2, Ny = 1152 (round 4, one-wave
  // columns): 4 columns 24.2 us, 3 columns (768 blocks = 3 per CU, 24-byte pieces) 26.7, 9 columns (one block per CU) 25.9.
  // Two-wave columns (Ny = 2304: 4096-row images): with the round-5 kernels (packed arithmetic, linear LDS indices: 110
  // registers, two blocks of 512 threads per CU) 4 columns 74.8 against 87.0 us for 2 (c6; round 4, 139 registers: 104
  // against 91).  The generic two-wave kernel (168 registers) keeps 2.
  // One-wave columns of 2048-row images (Ny = 1152), round 5: 8 columns per block (64-byte pieces, two blocks of 512 threads
  // per CU) 117 against 133 us for 4 per launch over 8 observations (c3 through the FFT path).
  const bool static_two_wave = lanes == 128 && fy.n == 3 && fy.r[0] == 16 && fy.r[1] == 16 && (fy.r[2] == 9 || fy.r[2] == 8);
  const bool static_1152 = lanes == 64 && fy.n == 3 && fy.r[0] == 16 && fy.r[1] == 8 && fy.r[2] == 9;
  int cb = static_1152 ? 8 : lanes == 64 || static_two_wave ? 4 : 2;
  while (cb > 2 && n.Nx % cb != 0) cb /= 2;
  const int ocb = opt_value(OPT_FFT_NATIVE, 1);  // (tuning: JD_FFT_NATIVE = 2 / 4 / 8 forces the columns per block)
  if ((ocb == 2 || ocb == 4 || ocb == 8) && (size_t)ocb * per_col <= 160 * 1024 && n.Nx % ocb == 0 && ocb * lanes <= 1024) cb = ocb;
  a.groups = n.Nx / cb;
  // the kernel of the schedule: compile-time forms for the lengths of the usual image sizes (default columns per block),
  // the generic form for everything else
  using Kernel = void (*)(ColsArgs);
  struct Entry {
    int lanes, cb, r0, r1, r2;
    Kernel kernel;
  };
  static Entry table[] = {
      {64, 4, 16, 8, 9, fftn_cols_kernel<64, 4, 16, 8, 9>},    // 1152: 2048-row images, PSFs up to 129 rows
      {64, 8, 16, 8, 9, fftn_cols_kernel<64, 8, 16, 8, 9>},    // (default there: 64-byte pieces, two blocks of 512 threads per CU)
      {128, 2, 16, 16, 9, fftn_cols_kernel<128, 2, 16, 16, 9>},  // 2304: 4096-row images
      {128, 4, 16, 16, 9, fftn_cols_kernel<128, 4, 16, 16, 9>},  // (default: four columns per block, 32-byte pieces)
      {128, 8, 16, 16, 9, fftn_cols_kernel<128, 8, 16, 16, 9>},  // (JD_FFT_NATIVE=8: one block of 1024 threads per CU, 64-byte pieces)
      {64, 4, 16, 8, 8, fftn_cols_kernel<64, 4, 16, 8, 8>},    // 1024
      {128, 2, 16, 16, 8, fftn_cols_kernel<128, 2, 16, 16, 8>},  // 2048
      {128, 4, 16, 16, 8, fftn_cols_kernel<128, 4, 16, 16, 8>},
      {64, 4, 8, 8, 9, fftn_cols_kernel<64, 4, 8, 8, 9>},      // 576: 1024-row images
      {64, 4, 8, 8, 8, fftn_cols_kernel<64, 4, 8, 8, 8>},      // 512
      {64, 0, 0, 0, 0, fftn_cols_kernel<64, 0, 0, 0, 0>},      // generic
      {128, 0, 0, 0, 0, fftn_cols_kernel<128, 0, 0, 0, 0>},
  };
  Entry* e = nullptr;
  for (Entry& t : table) {
    const bool is_static = t.r0 && fy.n == 3 && t.r0 == fy.r[0] && t.r1 == fy.r[1] && t.r2 == fy.r[2] && t.cb == cb;
    if (t.lanes == lanes && (is_static || !t.r0) && !e) e = &t;
  }
  int rc = lds_attr(reinterpret_cast<const void*>(e->kernel), cb * per_col);
  if (rc) return rc;
  ProfScope prof(JD_KERNEL_CMUL, stream);
  hipLaunchKernelGGL(e->kernel, dim3(((a.groups + 7) / 8) * 8 * (n_batch ? n_batch : 1)), dim3(lanes * cb), cb * per_col, stream, a);
  JD_LAUNCH_CHECK();
  return JD_OK;
}

// The row kernels by schedule: compile-time forms for the row lengths of the usual image sizes, the generic form
// otherwise.  Index into every table: row_schedule(f).
constexpr int N_ROW_SCHED = 6;
int fftn_cus() {  // compute units of the current device
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    n_cu = hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
  }
  return n_cu;
}
int row_schedule(const FftPasses& f, int Nx, int W, int blocks) {
  auto is = [&](int r0, int r1, int r2, int r3) {
    return f.n == (r3 ? 4 : 3) && f.r[0] == r0 && f.r[1] == r1 && f.r[2] == r2 && (!r3 || f.r[3] == r3);
  };
  if (is(16, 16, 9, 0)) return 1;  // 2304: 2048-column images
  if (is(8, 8, 8, 9)) return 2;    // 4608: 4096-column images
  if (is(16, 8, 9, 0)) return 3;   // 1152: 1024-column images
  // one wave per row where the launch has rows enough to fill the chip with waves (8 batched datasets at 512^2: 2592 row
  // pairs, 150 -> 126 us per step); a launch of few rows (one dataset: 384) is done sooner with four waves on each
  // (e0102, sequential: 0.240 against 0.263 ms per step)
  const int tiny = opt_value(OPT_FFT_TINY, GENERIC_TINY);
  if (Nx <= tiny && Nx <= GENERIC_TINY && W <= GENERIC_TINY && (blocks >= 3 * fftn_cus() || opt_is_set(OPT_FFT_TINY))) return 5;
  return Nx <= GENERIC_SMALL && W <= GENERIC_SMALL ? 4 : 0;
}
#define JD_ROW_KERNELS(NAME, ...)                                                                          \
  {NAME<__VA_ARGS__ 0, 0, 0, 0>, NAME<__VA_ARGS__ 16, 16, 9, 0>, NAME<__VA_ARGS__ 8, 8, 8, 9>, NAME<__VA_ARGS__ 16, 8, 9, 0>, \
   NAME<__VA_ARGS__ 0, 0, 0, 1>, NAME<__VA_ARGS__ 0, 0, 0, 2>}

template <class Args>
int launch_row_kernel(void (*const (&kernels)[N_ROW_SCHED])(Args), const FftNative& n, const Args& a,
                      int kernel_id, hipStream_t stream, int blocks = 0) {
  const int sched = row_schedule(a.f, n.Nx, n.W, blocks ? blocks : n.Hh);
  const size_t lds_rows = (size_t)2 * lp_size(n.Nx) * sizeof(float2);
  int rc = lds_attr(reinterpret_cast<const void*>(kernels[sched]), lds_rows);
  if (rc) return rc;
  ProfScope prof(kernel_id, stream);
  static constexpr int threads[N_ROW_SCHED] = {RowSched<0, 0, 0, 0>::T, RowSched<16, 16, 9, 0>::T, RowSched<8, 8, 8, 9>::T, RowSched<16, 8, 9, 0>::T,
                                               RowSched<0, 0, 0, 1>::T, RowSched<0, 0, 0, 2>::T};
  hipLaunchKernelGGL(kernels[sched], dim3(blocks ? blocks : n.Hh), dim3(threads[sched]), lds_rows, stream, a);
  JD_LAUNCH_CHECK();
  return JD_OK;
}

int launch_rows_fwd(const FftNative& n, const float* in, const float* in_scale, hipStream_t stream, const float* shift_xy = nullptr,
                    float shift_scale = 1.f, const FftBatch* batch = nullptr, int n_batch = 0, int d0 = 0) {
  static void (*const kernels[N_ROW_SCHED])(RowsFwdArgs) = JD_ROW_KERNELS(fftn_rows_fwd_kernel, );
  RowsFwdArgs a{};
  a.in = in, a.scale = in_scale, a.spec = n.spec, a.tw = n.tw_x, a.H = n.H, a.W = n.W, a.Hh = n.Hh, a.Nx = n.Nx, a.f = passes_of(n.Nx);
  a.shift_xy = shift_xy, a.shift_scale = shift_scale;
  a.batch = batch, a.n_batch = n_batch, a.d0 = d0;
  return launch_row_kernel(kernels, n, a, JD_KERNEL_FFT_R2C, stream, n_batch ? n.Hh * n_batch : 0);
}

int launch_rows_inv(const FftNative& n, float* out, const float* out_scale, int adjoint, float coef, int accumulate, hipStream_t stream,
                    const SepLossFold* fold = nullptr, const SepLossFold* fold2 = nullptr, const float2* work = nullptr) {
  static void (*const kernels_fwd[N_ROW_SCHED])(RowsInvArgs) = JD_ROW_KERNELS(fftn_rows_inv_kernel, false, );
  static void (*const kernels_adj[N_ROW_SCHED])(RowsInvArgs) = JD_ROW_KERNELS(fftn_rows_inv_kernel, true, );
  RowsInvArgs a{};
  a.work = work ? work : n.work, a.tw = n.tw_x, a.out = out, a.scale = out_scale, a.H = n.H, a.W = n.W, a.Hh = n.Hh, a.Nx = n.Nx, a.Ny = n.Ny;
  a.ra = adjoint ? n.kh - 1 - n.oy : n.oy, a.rb = adjoint ? n.oy : n.kh - 1 - n.oy;
  a.coef = coef, a.accumulate = accumulate, a.f = passes_of(n.Nx);
  if (fold && adjoint)
    a.fin_partials = fold->partials, a.fin_count = fold->count, a.fin_scale = fold->scale, a.fin_offset = fold->offset, a.fin_out = fold->out;
  if (fold && fold2 && adjoint) a.fin2_partials = fold2->partials, a.fin2_scale = fold2->scale, a.fin2_out = fold2->out;
  return adjoint ? launch_row_kernel(kernels_adj, n, a, JD_KERNEL_FFT_C2R, stream)
                 : launch_row_kernel(kernels_fwd, n, a, JD_KERNEL_FFT_C2R, stream);
}
}  // namespace

// out = conv_same(in * in_scale, psf)                       (adjoint == 0)
// out (+)= coef * out_scale * corr_same(in, psf)            (adjoint != 0; the transpose of the above)
int fftn_conv(const FftNative& n, const float* in, const float* in_scale, const float2* khat, float* out, const float* out_scale,
              int adjoint, float coef, int accumulate, hipStream_t stream) {
  int rc = launch_rows_fwd(n, in, in_scale, stream);
  if (rc) return rc;
  if ((rc = launch_cols(n, khat, adjoint, stream))) return rc;
  return launch_rows_inv(n, out, out_scale, adjoint, coef, accumulate, stream);
}

// The likelihood step of one dataset and one flux component in FIVE launches: rows(flux x exposure), columns(K^),
// rows^-1 + Poisson pass + rows(g), columns(conj K^), rows^-1 + adjoint epilogue.  partials[0 .. *n_partials): block
// sums of n - c log(n + eps);  grad (+)= coef * exposure * corr(g, psf);  *loss_out = loss_scale * sum(partials) + loss_offset.
int fftn_poisson_step(const FftNative& n, const float* flux, const float* exposure, const float2* khat, const float* background,
                      const float* counts, double* partials, int* n_partials, float eps, float inv_n, float* grad, float coef,
                      int accumulate, hipStream_t stream, double loss_scale, double loss_offset, float* loss_out) {
  int rc = launch_rows_fwd(n, flux, exposure, stream);
  if (rc) return rc;
  if ((rc = launch_cols(n, khat, 0, stream))) return rc;
  {
    static void (*const kernels[N_ROW_SCHED])(RowsPoissonArgs) = JD_ROW_KERNELS(fftn_rows_poisson_kernel, );
    RowsPoissonArgs a{};
    a.work = n.work, a.spec = n.spec, a.tw = n.tw_x, a.background = background, a.counts = counts, a.partials = partials;
    a.H = n.H, a.W = n.W, a.Hh = n.Hh, a.Nx = n.Nx, a.Ny = n.Ny, a.ra = n.oy, a.rb = n.kh - 1 - n.oy;
    a.eps = eps, a.inv_n = inv_n, a.f = passes_of(n.Nx);
    if ((rc = launch_row_kernel(kernels, n, a, JD_KERNEL_POISSON_FUSED, stream))) return rc;
  }
  *n_partials = n.Hh;
  if ((rc = launch_cols(n, khat, 1, stream))) return rc;
  // the loss = loss_scale * sum(partials) + loss_offset, by block 0 of the last launch
  const SepLossFold fold{partials, n.Hh, loss_scale, loss_offset, loss_out};
  return launch_rows_inv(n, grad, exposure, 1, coef, accumulate, stream, &fold);
}

// The same with up-sampling U = 2 or 4 (models/npred.py:181-184) and an optional background norm (:236-239): rows(flux x
// exposure; `shift_xy` != null: of the bilinearly shifted flux), columns, rows^-1 x U + pool + Poisson pass + rows(up-sampled
// g), columns(conj), rows^-1 + adjoint epilogue
// into `target` (= coef * exposure * corr, overwritten or accumulated).  The loss and, if wanted, d loss / d log norm =
// norm_grad_scale * sum(g * background) are finalised by blocks 0 and 1 of the last launch.
bool fftn_pooled_supported(const FftNative& n, int upsampling) {
  return upsampling >= 2 && upsampling <= 4 && n.W % upsampling == 0 && n.H % (2 * upsampling) == 0 && n.Hh / upsampling >= 2 &&
         n.W <= 4 * ROW_THREADS * 5;
}

namespace {
// U where the column passes around the pooled launch exchange row-group sums / single rows with it (ColsArgs::pool_out):
// the groups must line up in the image half, in the spill rows behind it (Hh + y) and in those in front (Ny - Hh + y)
int pooled_column_io(const FftNative& n, int upsampling) {
  const int U = upsampling, ra = n.oy, rb = n.kh - 1 - n.oy;
  const bool ok = U > 1 && n.Hh % U == 0 && n.Ny % U == 0 && n.Hh / U >= (ra + U - 1) / U + (rb + U - 1) / U + 1 &&
                  opt_value(OPT_FFT_POOL_IO, 1) != 0;  // (JD_FFT_POOL_IO=0: full rows both ways, as until the middle of round 5)
  return ok ? U : 1;
}

int launch_rows_pooled(const FftNative& n, int upsampling, const RowsPooledArgs& a, hipStream_t stream, int blocks) {
  static void (*const kernels2[N_ROW_SCHED])(RowsPooledArgs) = JD_ROW_KERNELS(fftn_rows_pooled_kernel, 2, );
  static void (*const kernels3[N_ROW_SCHED])(RowsPooledArgs) = JD_ROW_KERNELS(fftn_rows_pooled_kernel, 3, );
  static void (*const kernels4[N_ROW_SCHED])(RowsPooledArgs) = JD_ROW_KERNELS(fftn_rows_pooled_kernel, 4, );
  return upsampling == 2   ? launch_row_kernel(kernels2, n, a, JD_KERNEL_POISSON_FUSED, stream, blocks)
         : upsampling == 3 ? launch_row_kernel(kernels3, n, a, JD_KERNEL_POISSON_FUSED, stream, blocks)
                           : launch_row_kernel(kernels4, n, a, JD_KERNEL_POISSON_FUSED, stream, blocks);
}
}  // namespace

int fftn_poisson_step_pooled(const FftNative& n, int upsampling, const float* flux, const float* exposure, const float2* khat,
                             const float* background, const float* counts, const float* log_bkg_norm, double* partials,
                             double* partials_b, float eps, float inv_n, float* target, float coef, int accumulate,
                             hipStream_t stream, double loss_scale, double loss_offset, float* loss_out, double norm_grad_scale,
                             float* norm_grad_out, const float* shift_xy, float shift_scale) {
  int rc = launch_rows_fwd(n, flux, exposure, stream, shift_xy, shift_scale);
  if (rc) return rc;
  const int pool = pooled_column_io(n, upsampling);
  if ((rc = launch_cols(n, khat, 0, stream, nullptr, 0, 0, pool))) return rc;
  {
    RowsPooledArgs a{};
    a.work = n.work, a.spec = n.spec, a.tw = n.tw_x, a.background = background, a.counts = counts, a.log_bkg_norm = log_bkg_norm;
    a.partials = partials, a.partials_b = norm_grad_out ? partials_b : nullptr;
    a.H = n.H, a.W = n.W, a.Hh = n.Hh, a.Nx = n.Nx, a.Ny = n.Ny, a.ra = n.oy, a.rb = n.kh - 1 - n.oy;
    a.pooled_io = pool > 1;
    a.eps = eps, a.inv_n = inv_n, a.f = passes_of(n.Nx);
    rc = launch_rows_pooled(n, upsampling, a, stream, n.Hh / upsampling);
    if (rc) return rc;
  }
  if ((rc = launch_cols(n, khat, 1, stream, nullptr, 0, 0, pool))) return rc;
  const SepLossFold fold{partials, n.Hh / upsampling, loss_scale, loss_offset, loss_out};
  const SepLossFold fold2{partials_b, n.Hh / upsampling, norm_grad_scale, 0.0, norm_grad_out};
  return launch_rows_inv(n, target, exposure, 1, coef, accumulate, stream, &fold, norm_grad_out ? &fold2 : nullptr);
}

// The likelihood steps of `nd` datasets of ONE flux image in five launches (every launch covers all datasets; the last one
// adds the datasets' gradients in order inside its blocks): the same sums, bit for bit, as fftn_poisson_step called for
// the datasets one after the other with `accumulate` from the second on.  batch_dev (device memory): exposure, khat,
// background, counts, spec, work, loss_out, loss_offset per dataset.  partials: nd * Hh doubles.
int fftn_poisson_step_batch(const FftNative& n, int nd, const FftBatch* batch_dev, const float* flux, double* partials, float eps,
                            float inv_n, float* grad, float coef, int accumulate, hipStream_t stream, double loss_scale) {
  if (nd < 1 || nd > FFT_MAX_BATCH) return fail(JD_ERR_INVALID, "native FFT batch: %d datasets not in [1, %d]", nd, FFT_MAX_BATCH);
  int rc = launch_rows_fwd(n, flux, nullptr, stream, nullptr, 1.f, batch_dev, nd);
  if (rc) return rc;
  if ((rc = launch_cols(n, nullptr, 0, stream, batch_dev, nd))) return rc;
  {
    static void (*const kernels[N_ROW_SCHED])(RowsPoissonArgs) = JD_ROW_KERNELS(fftn_rows_poisson_kernel, );
    RowsPoissonArgs a{};
    a.tw = n.tw_x, a.partials = partials;
    a.H = n.H, a.W = n.W, a.Hh = n.Hh, a.Nx = n.Nx, a.Ny = n.Ny, a.ra = n.oy, a.rb = n.kh - 1 - n.oy;
    a.eps = eps, a.inv_n = inv_n, a.f = passes_of(n.Nx), a.batch = batch_dev, a.n_batch = nd;
    if ((rc = launch_row_kernel(kernels, n, a, JD_KERNEL_POISSON_FUSED, stream, n.Hh * nd))) return rc;
  }
  if ((rc = launch_cols(n, nullptr, 1, stream, batch_dev, nd))) return rc;
  static void (*const kernels[N_ROW_SCHED])(RowsInvArgs) = JD_ROW_KERNELS(fftn_rows_inv_batch_kernel, );
  RowsInvArgs a{};
  a.tw = n.tw_x, a.out = grad, a.H = n.H, a.W = n.W, a.Hh = n.Hh, a.Nx = n.Nx, a.Ny = n.Ny;
  a.ra = n.kh - 1 - n.oy, a.rb = n.oy;
  a.coef = coef, a.accumulate = accumulate, a.f = passes_of(n.Nx), a.batch = batch_dev, a.n_batch = nd;
  a.fin_partials = partials, a.fin_count = n.Hh, a.fin_scale = loss_scale;
  return launch_row_kernel(kernels, n, a, JD_KERNEL_FFT_C2R, stream);
}

// The likelihood steps of `nd` datasets of one flux image with up-sampling U = 2 / 4 and, per dataset, an optional
// calibration (shift_xy / log_bkg_norm entries of the table, nullable): rows (with the dataset's shift), columns, pooled
// middle launch and the adjoint's column pass each cover ALL datasets; the tail runs per dataset in order -- rows^-1 +
// adjoint epilogue (into `gshift` where the dataset has a shift, else accumulated into `grad`; its blocks 0 / 1 finalise
// the dataset's loss and d loss / d log norm), then the transposed shift (+ its two partial sums) and their finalize -- so
// the gradient is summed in dataset order: the per-dataset calls' results, bit for bit.
// partials / partials_b: nd * Hh / U doubles each; partials_shift: nd * 2 * shift_bwd_max_blocks(H, W) doubles.
int fftn_poisson_step_pooled_batch(const FftNative& n, int upsampling, int nd, const FftBatch* batch_dev, const FftBatch& host,
                                   const float* flux, double* partials, double* partials_b, float eps, float inv_n, float* grad,
                                   double* partials_shift, float coef, int accumulate, hipStream_t stream, double loss_scale,
                                   double norm_grad_scale, int sequential) {
  if (nd < 1 || nd > FFT_MAX_BATCH) return fail(JD_ERR_INVALID, "native FFT batch: %d datasets not in [1, %d]", nd, FFT_MAX_BATCH);
  if (!fftn_pooled_supported(n, upsampling)) return fail(JD_ERR_INVALID, "native FFT batch: up-sampling %d not supported", upsampling);
  const int per = n.Hh / upsampling;
  // sequential (large images, where a dataset's launches already fill the chip in several rounds of blocks and its 160 MB
  // of work arrays stay in the 256 MB last-level cache from launch to launch): the five FFT launches dataset by dataset
  // (blocks of dataset d0 only; the table may give every dataset the SAME work arrays), then the tail over all datasets
  const int groups = sequential ? nd : 1, per_launch = sequential ? 1 : nd;
  for (int gi = 0; gi < groups; ++gi) {
    const int d0 = sequential ? gi : 0;
    int rc = launch_rows_fwd(n, flux, nullptr, stream, nullptr, (float)upsampling, batch_dev, per_launch, d0);
    if (rc) return rc;
    const int pool = pooled_column_io(n, upsampling);
    if ((rc = launch_cols(n, nullptr, 0, stream, batch_dev, per_launch, d0, pool))) return rc;
    {
      RowsPooledArgs a{};
      a.tw = n.tw_x, a.partials = partials, a.partials_b = partials_b;
      a.H = n.H, a.W = n.W, a.Hh = n.Hh, a.Nx = n.Nx, a.Ny = n.Ny, a.ra = n.oy, a.rb = n.kh - 1 - n.oy;
      a.pooled_io = pool > 1;
      a.eps = eps, a.inv_n = inv_n, a.f = passes_of(n.Nx), a.batch = batch_dev, a.n_batch = per_launch, a.d0 = d0;
      rc = launch_rows_pooled(n, upsampling, a, stream, per * per_launch);
      if (rc) return rc;
    }
    if ((rc = launch_cols(n, nullptr, 1, stream, batch_dev, per_launch, d0, pool))) return rc;
    // rows^-1 + adjoint epilogue into every dataset's own image (the blocks of row pairs 0 / 1 finalise its loss / norm
    // gradient)
    {
      static void (*const kernels_adj[N_ROW_SCHED])(RowsInvArgs) = JD_ROW_KERNELS(fftn_rows_inv_kernel, true, );
      RowsInvArgs a{};
      a.tw = n.tw_x, a.H = n.H, a.W = n.W, a.Hh = n.Hh, a.Nx = n.Nx, a.Ny = n.Ny;
      a.ra = n.kh - 1 - n.oy, a.rb = n.oy;
      a.coef = coef, a.accumulate = 0, a.f = passes_of(n.Nx), a.batch = batch_dev, a.n_batch = per_launch, a.d0 = d0;
      a.fin_partials = partials, a.fin_count = per, a.fin_scale = loss_scale;
      a.fin2_partials = partials_b, a.fin2_scale = norm_grad_scale;
      if ((rc = launch_row_kernel(kernels_adj, n, a, JD_KERNEL_FFT_C2R, stream, n.Hh * per_launch))) return rc;
    }
  }
  // the tail over all datasets: ONE transposed-shift launch that adds the datasets up in order (a dataset without a shift:
  // its image as it is), one launch for the shift gradients
  int rc;
  const size_t shift_stride = (size_t)2 * shift_bwd_max_blocks(n.H, n.W);
  int shift_blocks = 0;
  if ((rc = launch_shift_bwd_batch(flux, batch_dev, nd, grad, accumulate, n.H, n.W, (float)upsampling, partials_shift, shift_stride,
                                   &shift_blocks, stream)))
    return rc;
  float* shift_out[FFT_MAX_BATCH] = {nullptr};
  bool any = false;
  for (int d = 0; d < nd; ++d) shift_out[d] = host.shift_xy[d] ? host.grad_shift_xy[d] : nullptr, any = any || shift_out[d];
  return any ? launch_finalize_multi_batch(partials_shift, shift_stride, shift_blocks, nd, shift_out, stream) : JD_OK;
}

}  // namespace jd
