// GMM patch prior on gfx950 (CDNA4).
//
// Forward: for every overlapping 8x8 patch x (mean subtracted) and every mixture component k
//     y_k = x^T P_k - m_k ,  q_k = sum_j w_j y_kj^2 ,  l_k = c_k - q_k / 2 ,  v = max_k l_k | logsumexp_k l_k
// (jolideco/priors/patches/gmm.py:262-281, priors/patches/core.py:189-246).  This is a dense
// contraction Y^T = P'^T X^T with M = 64*K whitened coordinates, N = patches, depth 64, i.e.
// 2*64*64 flop per (patch, component): FLOP-bound on the fp32 roof.  It runs on the exact-fp32
// matrix cores (v_mfma_f32_32x32x2_f32: bit-for-bit an fmaf chain, same peak as the vector ALU
// but one operand VGPR per MFMA and the VALU left free for the epilogue):
//   * A operand = P'_k fragments (sqrt(w_j) folded into column j, fragment order prepared once on
//     the host) streamed from L2 with 16 B/lane loads, register double-buffered;
//   * B operand = the wave's patches, resident in VGPRs for the whole kernel (T tiles of 32);
//   * the accumulator is initialised with -m'_k so the mean shift costs nothing;
//   * C layout puts the patch on the lane and the whitened coordinate j in the registers, so
//     sum_j y_j^2 is an in-lane sum + one cross-half shuffle; (Np, K) never leaves the CU.
// Backward (max mode): only the arg-max component contributes; a second, small kernel recomputes
// y for that component and applies P' once more, the overlap-add is done race-free and in a fixed
// order by a gather pass (every pixel sums its <= 4 patch contributions).
#include <cmath>
#include <cstdlib>
#include <vector>

#include "jd_common.h"
#include "kernels.h"

namespace jd {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int P = 8;    // patch edge
constexpr int D = 64;   // features per patch
constexpr int FRAG_FLOATS = 2048;  // one (k, row-block) A fragment: 8 x 64 lanes x float4

enum { MODE_MAX = 0, MODE_LSE = 1, MODE_DENSE = 2 };

struct GmmFwdArgs {
  const float* flux;     // (H, W) image  | MODE_DENSE: (n, 64) explicit patches
  const float* pfrag;    // K * 2 * FRAG_FLOATS
  const float* mfrag;    // K * 2 * 2 * 16   (negated m')
  const float* const_k;  // K
  int K, H, W, stride, nPx, shift_y, shift_x;
  int n_begin, n_end;    // linear patch index range (row-major over the patch grid)
  int32_t* argmax_out;   // nullable (MODE_MAX)
  float* value_patch;    // nullable: per patch v (MODE_LSE backward needs it) | MODE_DENSE: (n, K) out
  double* partials;      // one per wave
};

__device__ __forceinline__ int wrap(int v, int n) {
  v %= n;
  return v < 0 ? v + n : v;
}

template <int T, int MODE>
__global__ __launch_bounds__(256) void gmm_fwd_kernel(GmmFwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int h = lane >> 5, c = lane & 31;
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int base = a.n_begin + wave_global * (32 * T);
  if (base >= a.n_end) return;  // whole wave idle (wave-uniform)

  // ---- B operand: T tiles of 32 patches; lane (h, c) keeps pixels 32h .. 32h+31 of patch c ----
  float x[T][32];
  bool ok[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int n = base + 32 * t + c;
    const bool valid = n < a.n_end;
    float sum = 0.f;
    bool sel = true;
    if (MODE == MODE_DENSE) {
#pragma unroll
      for (int s = 0; s < 32; ++s) x[t][s] = valid ? a.flux[(size_t)n * D + 32 * h + s] : 0.f;
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
          x[t][8 * r + cc] = v;
          sum += v;
          sel = sel && (v > -1e5f);  // patches/core.py:215
        }
      }
      sum += __shfl_xor(sum, 32, 64);
      const float mean = sum * (1.f / 64.f);  // SubtractMeanPatchNorm, utils/norms.py:100-103
#pragma unroll
      for (int s = 0; s < 32; ++s) x[t][s] -= mean;
      const int sel_other = __shfl_xor((int)sel, 32, 64);
      sel = sel && (sel_other != 0);
    }
    ok[t] = valid && sel;
  }

  float best[T], aux[T];  // MODE_MAX: best value | MODE_LSE: running max, running sum of exp
  int arg[T];
#pragma unroll
  for (int t = 0; t < T; ++t) best[t] = -INFINITY, aux[t] = 0.f, arg[t] = 0;

  const float4* pf = reinterpret_cast<const float4*>(a.pfrag) + lane;
  const float4* mf = reinterpret_cast<const float4*>(a.mfrag) + h * 4;

  float4 A0[8], A1[8];
#pragma unroll
  for (int qd = 0; qd < 8; ++qd) A0[qd] = pf[qd * 64];

  for (int k = 0; k < a.K; ++k) {
    const float4* pk = pf + (size_t)k * (2 * FRAG_FLOATS / 4);
    const float4* mk = mf + (size_t)k * 16;  // 2 rb * 2 h * 4 float4
    float4 m0[4], m1[4];
#pragma unroll
    for (int qd = 0; qd < 8; ++qd) A1[qd] = pk[FRAG_FLOATS / 4 + qd * 64];
#pragma unroll
    for (int i = 0; i < 4; ++i) m0[i] = mk[i], m1[i] = mk[8 + i];

    float q[T];
    // ---- row block 0 (whitened coordinates j = 0..31) ----
#pragma unroll
    for (int t = 0; t < T; ++t) {
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[4 * i] = m0[i].x, acc[4 * i + 1] = m0[i].y, acc[4 * i + 2] = m0[i].z, acc[4 * i + 3] = m0[i].w;
#pragma unroll
      for (int s = 0; s < 32; ++s) {
        const float av = (s & 3) == 0 ? A0[s >> 2].x : (s & 3) == 1 ? A0[s >> 2].y : (s & 3) == 2 ? A0[s >> 2].z : A0[s >> 2].w;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, x[t][s], acc, 0, 0, 0);
      }
      float qq = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) qq = fmaf(acc[i], acc[i], qq);
      q[t] = qq;
    }
    // prefetch the next component's first fragment while row block 1 computes
    if (k + 1 < a.K) {
#pragma unroll
      for (int qd = 0; qd < 8; ++qd) A0[qd] = pk[2 * FRAG_FLOATS / 4 + qd * 64];
    }
    // ---- row block 1 (j = 32..63) ----
#pragma unroll
    for (int t = 0; t < T; ++t) {
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[4 * i] = m1[i].x, acc[4 * i + 1] = m1[i].y, acc[4 * i + 2] = m1[i].z, acc[4 * i + 3] = m1[i].w;
#pragma unroll
      for (int s = 0; s < 32; ++s) {
        const float av = (s & 3) == 0 ? A1[s >> 2].x : (s & 3) == 1 ? A1[s >> 2].y : (s & 3) == 2 ? A1[s >> 2].z : A1[s >> 2].w;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, x[t][s], acc, 0, 0, 0);
      }
      float qq = q[t];
#pragma unroll
      for (int i = 0; i < 16; ++i) qq = fmaf(acc[i], acc[i], qq);
      q[t] = qq;
    }
    // ---- per component epilogue: l = c_k - q/2 (gmm.py:276-281), then max / online logsumexp ----
    const float ck = a.const_k[k];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const float qf = q[t] + __shfl_xor(q[t], 32, 64);
      const float l = fmaf(-0.5f, qf, ck);
      if (MODE == MODE_MAX) {
        if (l > best[t]) best[t] = l, arg[t] = k;
      } else if (MODE == MODE_LSE) {
        if (l > best[t]) {
          aux[t] = aux[t] * expf(best[t] - l) + 1.f;
          best[t] = l;
        } else {
          aux[t] += expf(l - best[t]);
        }
      } else {
        const int n = base + 32 * t + c;
        if (h == 0 && n < a.n_end) a.value_patch[(size_t)n * a.K + k] = l;
      }
    }
  }

  if (MODE == MODE_DENSE) return;

  double local = 0.0;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int n = base + 32 * t + c;
    float v = best[t];
    if (MODE == MODE_LSE) v = best[t] + logf(aux[t]);
    if (h == 0 && n < a.n_end) {
      if (MODE == MODE_MAX && a.argmax_out) a.argmax_out[n] = ok[t] ? arg[t] : -1;
      if (a.value_patch) a.value_patch[n] = ok[t] ? v : NAN;
      if (ok[t]) local += (double)v;
    }
  }
  local = wave_sum(local);
  if (lane == 0) a.partials[wave_global] = local;
}

// ------------------------------------------------------------------------------------------
// Backward, max mode: per patch  gamma = -P'_k* (xbar^T P'_k* - m'_k*),  gbar = gamma - mean(gamma).
// Every patch uses the matrix of ITS arg-max component, so the patches are first bucketed by
// component (counting sort: LDS histograms + one global atomic per bin and block; the order inside a
// bucket does not influence any result); buckets are padded to 32 slots.  One wave then takes a
// 32-slot group, i.e. 32 patches that share P'_k, and runs both products on the matrix cores:
//   Y^T = P'^T Xbar^T - m'      (as in the forward kernel)
//   G^T = P' Y^T                (the Y accumulators ARE the B operand: lane (h, c) holds Y[j][c] for 32
//                                values of j, and the A fragments of this product are laid out on the
//                                host in exactly that j order, so no lane movement / LDS is needed)
// 128 MFMAs per 32 patches instead of 2 x 16 KB of matrix reads per patch.
// ------------------------------------------------------------------------------------------
struct GmmBucketArgs {
  const int32_t* argmax;  // global patch index -> component or -1
  int n_begin, n_end, K;
  int* counts;    // K      (zeroed by the caller)
  int* cursor;    // K      (zeroed by the caller)
  int* offsets;   // K + 1  exclusive scan of the padded counts; offsets[K] = total slots
  int32_t* order; // slot -> global patch index, -1 for padding (pre-filled with -1 by the caller)
  float* gpatch;  // rows of filtered patches (argmax < 0) are zeroed here
};

constexpr int BUCKET_CHUNK = 1024;  // patches per block (4 per thread)
constexpr int BUCKET_MAX_K = 4096;  // LDS histogram capacity

__global__ __launch_bounds__(256) void gmm_bucket_count_kernel(GmmBucketArgs a) {
  extern __shared__ int hist[];
  for (int k = threadIdx.x; k < a.K; k += 256) hist[k] = 0;
  __syncthreads();
  const int base = a.n_begin + blockIdx.x * BUCKET_CHUNK;
#pragma unroll
  for (int i = 0; i < BUCKET_CHUNK / 256; ++i) {
    const int n = base + i * 256 + threadIdx.x;
    if (n < a.n_end) {
      const int k = a.argmax[n];
      if (k >= 0) atomicAdd(&hist[k], 1);
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < a.K; k += 256)
    if (hist[k]) atomicAdd(&a.counts[k], hist[k]);
}

__global__ __launch_bounds__(256) void gmm_bucket_scan_kernel(GmmBucketArgs a) {
  // exclusive scan of the padded bucket sizes: thread t owns a contiguous segment of bins,
  // the 256 segment sums are scanned in LDS (Hillis-Steele)
  __shared__ int part[2][256];
  const int seg = (a.K + 255) / 256;
  const int k0 = threadIdx.x * seg;
  int local = 0;
  for (int k = k0; k < k0 + seg && k < a.K; ++k) local += (a.counts[k] + 31) & ~31;
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
  int total = part[cur][threadIdx.x] - local;  // exclusive prefix of this thread's segment
  for (int k = k0; k < k0 + seg && k < a.K; ++k) {
    a.offsets[k] = total;
    total += (a.counts[k] + 31) & ~31;
  }
  if (threadIdx.x == 255) a.offsets[a.K] = part[cur][255];
}

__global__ __launch_bounds__(256) void gmm_bucket_scatter_kernel(GmmBucketArgs a) {
  extern __shared__ int hist[];  // [0, K): block-local counts, then the block's base inside each bucket
  for (int k = threadIdx.x; k < a.K; k += 256) hist[k] = 0;
  __syncthreads();
  const int base = a.n_begin + blockIdx.x * BUCKET_CHUNK;
  int kk[BUCKET_CHUNK / 256], rank[BUCKET_CHUNK / 256];
#pragma unroll
  for (int i = 0; i < BUCKET_CHUNK / 256; ++i) {
    const int n = base + i * 256 + threadIdx.x;
    kk[i] = -2;
    if (n < a.n_end) {
      kk[i] = a.argmax[n];
      if (kk[i] >= 0) {
        rank[i] = atomicAdd(&hist[kk[i]], 1);
      } else {  // filtered patch (patches/core.py:215-216): no gradient
        float4* row = reinterpret_cast<float4*>(a.gpatch + (size_t)(n - a.n_begin) * D);
        for (int q = 0; q < D / 4; ++q) row[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < a.K; k += 256) {
    const int c = hist[k];
    hist[k] = c ? a.offsets[k] + atomicAdd(&a.cursor[k], c) : 0;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < BUCKET_CHUNK / 256; ++i)
    if (kk[i] >= 0) a.order[hist[kk[i]] + rank[i]] = base + i * 256 + threadIdx.x;
}

struct GmmBwdArgs {
  const float* flux;
  const float* pfrag;  // as in the forward kernel
  const float* mfrag;
  const float* gfrag;  // K * 2 * FRAG_FLOATS: A fragments of the second product
  const int32_t* argmax;
  const int32_t* order;
  const int* offsets;  // offsets[K] = total slots
  float* gpatch;       // (n_end - n_begin) * 64
  int K, H, W, stride, nPx, shift_y, shift_x, n_begin, n_end;
};

__global__ __launch_bounds__(256) void gmm_bwd_max_kernel(GmmBwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int h = lane >> 5, c = lane & 31;
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int n_waves = gridDim.x * 4;
  const int n_groups = a.offsets[a.K] >> 5;
  for (int g = wave_global; g < n_groups; g += n_waves) {
    const int n = a.order[32 * g + c];
    const bool valid = n >= 0;
    // slot 0 of a group is always occupied (padding sits at the end of a bucket)
    const int k = __builtin_amdgcn_readfirstlane(a.argmax[__builtin_amdgcn_readfirstlane(n)]);

    // ---- B operand: pixels 32h .. 32h+31 of patch c, mean subtracted --------------------------
    float x[32];
    {
      const int py = valid ? n / a.nPx : 0, px = valid ? n % a.nPx : 0;
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int yy = wrap(py * a.stride + 4 * h + r - a.shift_y, a.H);
        const float* row = a.flux + (size_t)yy * a.W;
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
          const int xx = wrap(px * a.stride + cc - a.shift_x, a.W);
          const float v = valid ? row[xx] : 0.f;
          x[8 * r + cc] = v;
          sum += v;
        }
      }
      sum += __shfl_xor(sum, 32, 64);
      const float mean = sum * (1.f / 64.f);
#pragma unroll
      for (int s = 0; s < 32; ++s) x[s] -= mean;
    }

    // ---- Y^T = P'^T Xbar^T - m' ------------------------------------------------------------------
    f32x16 y[2];
    {
      const float4* pk = reinterpret_cast<const float4*>(a.pfrag) + (size_t)k * (2 * FRAG_FLOATS / 4) + lane;
      const float4* mk = reinterpret_cast<const float4*>(a.mfrag) + (size_t)k * 16 + h * 4;
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        float4 A[8];
#pragma unroll
        for (int qd = 0; qd < 8; ++qd) A[qd] = pk[rb * (FRAG_FLOATS / 4) + qd * 64];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 m = mk[rb * 8 + i];
          y[rb][4 * i] = m.x, y[rb][4 * i + 1] = m.y, y[rb][4 * i + 2] = m.z, y[rb][4 * i + 3] = m.w;
        }
#pragma unroll
        for (int s = 0; s < 32; ++s) {
          const float av = (s & 3) == 0 ? A[s >> 2].x : (s & 3) == 1 ? A[s >> 2].y : (s & 3) == 2 ? A[s >> 2].z : A[s >> 2].w;
          y[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, x[s], y[rb], 0, 0, 0);
        }
      }
    }

    // ---- G^T = P' Y^T : k-step s feeds lane (h, c) value y[s >> 4][s & 15] ------------------------
    f32x16 gacc[2];
    {
      const float4* gk = reinterpret_cast<const float4*>(a.gfrag) + (size_t)k * (2 * FRAG_FLOATS / 4) + lane;
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        float4 A[8];
#pragma unroll
        for (int qd = 0; qd < 8; ++qd) A[qd] = gk[pb * (FRAG_FLOATS / 4) + qd * 64];
#pragma unroll
        for (int i = 0; i < 16; ++i) gacc[pb][i] = 0.f;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
          const float av = (s & 3) == 0 ? A[s >> 2].x : (s & 3) == 1 ? A[s >> 2].y : (s & 3) == 2 ? A[s >> 2].z : A[s >> 2].w;
          gacc[pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, y[s >> 4][s & 15], gacc[pb], 0, 0, 0);
        }
      }
    }

    // ---- gamma = -G, subtract its mean over the 64 pixels (adjoint of the mean subtraction) -------
    float sum = 0.f;
#pragma unroll
    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
      for (int i = 0; i < 16; ++i) sum += gacc[pb][i];
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * (1.f / 64.f);
    if (valid) {
      // lane (h, c) holds pixels 32 pb + 8 q + 4 h + (0..3) of its patch in gacc[pb][4 q .. 4 q + 3]
      float4* out = reinterpret_cast<float4*>(a.gpatch + (size_t)(n - a.n_begin) * D);
#pragma unroll
      for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          out[8 * pb + 2 * q + h] = make_float4(mean - gacc[pb][4 * q], mean - gacc[pb][4 * q + 1],
                                                mean - gacc[pb][4 * q + 2], mean - gacc[pb][4 * q + 3]);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Overlap-add gather: every pixel of the rolled frame sums the contributions of the patches that
// cover it in a fixed order (no float atomics), un-rolls and accumulates into grad.
// ------------------------------------------------------------------------------------------
struct GmmGatherArgs {
  const float* gpatch;
  float* grad;
  int H, W, stride, nPx, nPy, shift_y, shift_x, row_begin, row_end;  // patch-row shard
  int y_begin, y_end;                                                // rolled-frame pixel rows covered
  float coef;
};

__global__ __launch_bounds__(256) void gmm_gather_kernel(GmmGatherArgs a) {
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
  float sum = 0.f;
  bool any = false;
  for (int py = py_lo; py <= py_hi; ++py) {
    const int r = Y - py * a.stride;
    for (int px = px_lo; px <= px_hi; ++px) {
      const int cc = X - px * a.stride;
      const size_t n = (size_t)(py - a.row_begin) * a.nPx + px;
      sum += a.gpatch[n * D + r * P + cc];
      any = true;
    }
  }
  if (!any) return;
  const int yy = wrap(Y - a.shift_y, a.H), xx = wrap(X - a.shift_x, a.W);
  a.grad[(size_t)yy * a.W + xx] += a.coef * sum;
}

}  // namespace jd

// ==========================================================================================
struct jd_gmm {
  int K = 0;
  float* pfrag = nullptr;
  float* mfrag = nullptr;
  float* const_k = nullptr;
  float* gfrag = nullptr;
  int* bucket = nullptr;  // counts (K) | cursor (K) | offsets (K + 1)
  // workspaces (grown on demand)
  int32_t* argmax = nullptr;
  size_t argmax_cap = 0;
  int32_t* order = nullptr;
  size_t order_cap = 0;
  float* gpatch = nullptr;
  size_t gpatch_cap = 0;
  double* partials = nullptr;
  size_t partials_cap = 0;
  int n_cu = 256;
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

  std::vector<float> pfrag((size_t)K * 2 * FRAG_FLOATS), gfrag((size_t)K * 2 * FRAG_FLOATS), mfrag((size_t)K * 64),
      prow((size_t)D * D), mrow(D);
  double sw[D];
  for (int j = 0; j < D; ++j) sw[j] = std::sqrt((double)pixel_w[j]);
  for (int k = 0; k < K; ++k) {
    const float* Pk = prec_chol + (size_t)k * D * D;
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < D; ++j) {
        prow[(size_t)i * D + j] = (float)((double)Pk[i * D + j] * sw[j]);  // P'[i][j] = P[i][j] * sqrt(w_j)
      }
    for (int j = 0; j < D; ++j) mrow[j] = (float)((double)mu_prec[(size_t)k * D + j] * sw[j]);
    // MFMA A fragments: [k][rb][qd][lane][e] = P'[pixel 32h + 4qd + e][j = 32rb + c]
    for (int rb = 0; rb < 2; ++rb)
      for (int qd = 0; qd < 8; ++qd)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 4; ++e) {
            const int hh = lane >> 5, cc = lane & 31;
            const int pix = 32 * hh + 4 * qd + e, j = 32 * rb + cc;
            pfrag[(((size_t)(k * 2 + rb) * 8 + qd) * 64 + lane) * 4 + e] = prow[(size_t)pix * D + j];
          }
    // A fragments of the backward product G^T = P' Y^T: [k][pb][qd][lane][e], k-step s = 4 qd + e
    // pairs with the Y accumulator register (rb = s >> 4, i = s & 15) of lane half hh, which holds
    // j = 32 rb + (i & 3) + 8 (i >> 2) + 4 hh
    for (int pb = 0; pb < 2; ++pb)
      for (int qd = 0; qd < 8; ++qd)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 4; ++e) {
            const int hh = lane >> 5, cc = lane & 31, st = 4 * qd + e;
            const int rb = st >> 4, i = st & 15;
            const int j = 32 * rb + (i & 3) + 8 * (i >> 2) + 4 * hh, pix = 32 * pb + cc;
            gfrag[(((size_t)(k * 2 + pb) * 8 + qd) * 64 + lane) * 4 + e] = prow[(size_t)pix * D + j];
          }
    // accumulator init: [k][rb][h][i] = -m'[j = 32rb + (i&3) + 8(i>>2) + 4h]
    for (int rb = 0; rb < 2; ++rb)
      for (int hh = 0; hh < 2; ++hh)
        for (int i = 0; i < 16; ++i) {
          const int j = 32 * rb + (i & 3) + 8 * (i >> 2) + 4 * hh;
          mfrag[(((size_t)k * 2 + rb) * 2 + hh) * 16 + i] = -mrow[j];
        }
  }
  auto upload = [&](float** dst, const float* src, size_t n) -> int {
    JD_HIP(hipMalloc(dst, n * sizeof(float)));
    JD_HIP(hipMemcpy(*dst, src, n * sizeof(float), hipMemcpyHostToDevice));
    return JD_OK;
  };
  int rc;
  if ((rc = upload(&g->pfrag, pfrag.data(), pfrag.size())) || (rc = upload(&g->mfrag, mfrag.data(), mfrag.size())) ||
      (rc = upload(&g->const_k, const_k, K)) || (rc = upload(&g->gfrag, gfrag.data(), gfrag.size()))) {
    jd_gmm_destroy(g);
    return rc;
  }
  if (hipMalloc(&g->bucket, (size_t)(3 * K + 1) * sizeof(int)) != hipSuccess) {
    jd_gmm_destroy(g);
    return fail(JD_ERR_ALLOC, "jd_gmm_create: hipMalloc of the bucket counters failed");
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
  for (float* p : {g->pfrag, g->mfrag, g->const_k, g->gfrag, g->gpatch})
    if (p) (void)hipFree(p);
  if (g->argmax) (void)hipFree(g->argmax);
  if (g->order) (void)hipFree(g->order);
  if (g->bucket) (void)hipFree(g->bucket);
  if (g->partials) (void)hipFree(g->partials);
  delete g;
  return JD_OK;
}

// Pick the number of 32-patch tiles per wave so that the grid fills the 4 SIMDs of every CU.
static int pick_tiles(long n_patches, int n_cu) {
  if (const char* env = getenv("JD_GMM_TILES")) {  // tuning override: 1, 2 or 4
    const int t = atoi(env);
    if (t == 1 || t == 2 || t == 4) return t;
  }
  const long simds = (long)n_cu * 4;
  for (int t : {4, 2}) {
    const long waves = (n_patches + 32L * t - 1) / (32L * t);
    if (waves >= 2 * simds) return t;
  }
  return 1;
}

template <int MODE>
static int launch_fwd(const GmmFwdArgs& a, int tiles, hipStream_t s, int* n_waves_out) {
  const long n = a.n_end - a.n_begin;
  const long per_wave = 32L * tiles;
  const long waves = (n + per_wave - 1) / per_wave;
  const unsigned blocks = (unsigned)((waves + 3) / 4);
  *n_waves_out = (int)waves;
  ProfScope prof(JD_KERNEL_GMM_FWD, s);
  switch (tiles) {
    case 4: gmm_fwd_kernel<4, MODE><<<blocks, 256, 0, s>>>(a); break;
    case 2: gmm_fwd_kernel<2, MODE><<<blocks, 256, 0, s>>>(a); break;
    default: gmm_fwd_kernel<1, MODE><<<blocks, 256, 0, s>>>(a); break;
  }
  JD_LAUNCH_CHECK();
  return JD_OK;
}

extern "C" int jd_gmm_prior_fwd_bwd(jd_gmm* g, const float* flux, int H, int W, int stride, int shift_y,
                                    int shift_x, int patch_row_begin, int patch_row_end, int marginalize,
                                    float value_scale, float* value_out, int accumulate_value, float grad_coef,
                                    float* grad_flux_accum, int32_t* argmax_out, void* stream) {
  JD_REQUIRE(g && flux && value_out, "jd_gmm_prior_fwd_bwd: null argument");
  JD_REQUIRE(H >= P && W >= P, "jd_gmm_prior_fwd_bwd: image (%d, %d) smaller than a patch", H, W);
  JD_REQUIRE(stride >= 1 && stride <= P, "jd_gmm_prior_fwd_bwd: stride = %d not in [1, 8]", stride);
  JD_REQUIRE(!(marginalize && grad_flux_accum),
             "jd_gmm_prior_fwd_bwd: the gradient of the marginalized (logsumexp) prior is not implemented");
  const int nPy = (H - P) / stride + 1, nPx = (W - P) / stride + 1;
  JD_REQUIRE((long)nPy * nPx < (1L << 31), "jd_gmm_prior_fwd_bwd: too many patches");
  if (patch_row_end < 0) patch_row_end = nPy;
  JD_REQUIRE(patch_row_begin >= 0 && patch_row_begin <= patch_row_end && patch_row_end <= nPy,
             "jd_gmm_prior_fwd_bwd: patch row range [%d, %d) outside [0, %d]", patch_row_begin, patch_row_end, nPy);
  hipStream_t s = as_stream(stream);
  const int n_begin = patch_row_begin * nPx, n_end = patch_row_end * nPx;
  if (n_begin == n_end) {  // empty shard: contributes nothing
    if (!accumulate_value) JD_HIP(hipMemsetAsync(value_out, 0, sizeof(float), s));
    return JD_OK;
  }
  const long n = n_end - n_begin;
  const int tiles = pick_tiles(n, g->n_cu);
  int rc;
  if ((rc = grow(&g->partials, &g->partials_cap, (size_t)((n + 31) / 32 + 4)))) return rc;
  int32_t* arg = argmax_out;
  if (!arg && grad_flux_accum) {
    if ((rc = grow(&g->argmax, &g->argmax_cap, (size_t)nPy * nPx))) return rc;
    arg = g->argmax;
  }
  GmmFwdArgs a{};
  a.flux = flux, a.pfrag = g->pfrag, a.mfrag = g->mfrag, a.const_k = g->const_k;
  a.K = g->K, a.H = H, a.W = W, a.stride = stride, a.nPx = nPx, a.shift_y = shift_y, a.shift_x = shift_x;
  a.n_begin = n_begin, a.n_end = n_end, a.argmax_out = arg, a.value_patch = nullptr, a.partials = g->partials;
  int n_waves = 0;
  if (marginalize)
    rc = launch_fwd<MODE_LSE>(a, tiles, s, &n_waves);
  else
    rc = launch_fwd<MODE_MAX>(a, tiles, s, &n_waves);
  if (rc) return rc;
  if ((rc = launch_finalize_sum(g->partials, n_waves, (double)value_scale, 0.0, value_out, accumulate_value, s)))
    return rc;
  if (!grad_flux_accum) return JD_OK;

  if ((rc = grow(&g->gpatch, &g->gpatch_cap, (size_t)n * D))) return rc;
  const size_t slots_cap = (size_t)n + 32 * (size_t)g->K;
  if ((rc = grow(&g->order, &g->order_cap, slots_cap))) return rc;
  // ---- bucket the patches by arg-max component -------------------------------------------------
  GmmBucketArgs bk{};
  bk.argmax = arg, bk.n_begin = n_begin, bk.n_end = n_end, bk.K = g->K;
  bk.counts = g->bucket, bk.cursor = g->bucket + g->K, bk.offsets = g->bucket + 2 * g->K;
  bk.order = g->order, bk.gpatch = g->gpatch;
  JD_HIP(hipMemsetAsync(g->bucket, 0, (size_t)2 * g->K * sizeof(int), s));
  JD_HIP(hipMemsetAsync(g->order, 0xFF, slots_cap * sizeof(int32_t), s));
  const unsigned chunks = (unsigned)((n + BUCKET_CHUNK - 1) / BUCKET_CHUNK);
  const size_t hist_bytes = (size_t)g->K * sizeof(int);
  {
    ProfScope prof(JD_KERNEL_GMM_BWD, s);
    gmm_bucket_count_kernel<<<chunks, 256, hist_bytes, s>>>(bk);
    gmm_bucket_scan_kernel<<<1, 256, 0, s>>>(bk);
    gmm_bucket_scatter_kernel<<<chunks, 256, hist_bytes, s>>>(bk);
    GmmBwdArgs b{};
    b.flux = flux, b.pfrag = g->pfrag, b.mfrag = g->mfrag, b.gfrag = g->gfrag, b.argmax = arg, b.order = g->order;
    b.offsets = bk.offsets, b.gpatch = g->gpatch, b.K = g->K;
    b.H = H, b.W = W, b.stride = stride, b.nPx = nPx, b.shift_y = shift_y, b.shift_x = shift_x;
    b.n_begin = n_begin, b.n_end = n_end;
    long bwd_blocks = ((long)(slots_cap / 32) + 3) / 4;
    const long cap = (long)g->n_cu * 3;  // 3 blocks of 4 waves per CU: one wave per SIMD x 3
    if (bwd_blocks > cap) bwd_blocks = cap;
    gmm_bwd_max_kernel<<<(unsigned)bwd_blocks, 256, 0, s>>>(b);
  }
  JD_LAUNCH_CHECK();

  GmmGatherArgs ga{};
  ga.gpatch = g->gpatch, ga.grad = grad_flux_accum, ga.H = H, ga.W = W, ga.stride = stride, ga.nPx = nPx, ga.nPy = nPy;
  ga.shift_y = shift_y, ga.shift_x = shift_x, ga.row_begin = patch_row_begin, ga.row_end = patch_row_end;
  ga.y_begin = patch_row_begin * stride;
  ga.y_end = (patch_row_end - 1) * stride + P;
  ga.coef = grad_coef;
  dim3 grid((W + 255) / 256, ga.y_end - ga.y_begin);
  {
    ProfScope prof(JD_KERNEL_GMM_GATHER, s);
    gmm_gather_kernel<<<grid, 256, 0, s>>>(ga);
  }
  JD_LAUNCH_CHECK();
  return JD_OK;
}

extern "C" int jd_gmm_estimate_log_prob(jd_gmm* g, const float* x, int n, float* out, void* stream) {
  JD_REQUIRE(g && x && out && n > 0, "jd_gmm_estimate_log_prob: null argument or n <= 0");
  hipStream_t s = as_stream(stream);
  int rc;
  if ((rc = grow(&g->partials, &g->partials_cap, (size_t)((n + 31) / 32 + 4)))) return rc;
  GmmFwdArgs a{};
  a.flux = x, a.pfrag = g->pfrag, a.mfrag = g->mfrag, a.const_k = g->const_k;
  a.K = g->K, a.n_begin = 0, a.n_end = n, a.value_patch = out, a.partials = g->partials;
  int n_waves = 0;
  return launch_fwd<MODE_DENSE>(a, pick_tiles(n, g->n_cu), s, &n_waves);
}
