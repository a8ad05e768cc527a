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
#include "jd_common.h"
#include "kernels.h"

namespace jd {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int TILE = 64;  // outputs per block edge

struct DirectConvArgs {
  const float* in;         // (H, W)
  const float* in_scale;   // nullable, multiplied onto `in` while staging (forward: exposure)
  const float* afrag;      // kh * (KC/4) * 64 floats: Toeplitz fragments of the (possibly flipped) PSF
  float* out;              // (H, W)
  const float* out_scale;  // nullable, multiplied onto the result (adjoint: exposure)
  int H, W, kh;
  int oy;     // output row y reads input rows y + oy - dy
  int ox_in;  // first input column of the strip starting at x0 is x0 + ox_in  (= ox - (kw-1))
  float coef;
  int accumulate;
};

template <int KC>
__global__ __launch_bounds__(256) void direct_conv_kernel(DirectConvArgs a) {
  constexpr int STEPS = KC / 4;
  constexpr int COLS = 48 + KC;    // window columns: 64 outputs + KC - 16 halo
  constexpr int PITCH = COLS + 2;  // == 2 (mod 4): the 16 rows x 2 columns of a half-wave hit 32 banks
  extern __shared__ __attribute__((aligned(16))) float win[];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int x0 = blockIdx.x * TILE, y0 = blockIdx.y * TILE;
  const int rows = TILE - 1 + a.kh;
  const int yin0 = y0 + a.oy - (a.kh - 1);
  const int xin0 = x0 + a.ox_in;

  // ---- stage the (rows x COLS) input window, zero outside the image ------------------------
  // U independent loads per thread are issued before the first one is consumed (one latency per
  // batch instead of one per element)
  constexpr int U = 6;
  const int total = rows * COLS;
  for (int base = threadIdx.x; base < total; base += 256 * U) {
    float v[U], sc[U];
    int dst[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + 256 * u;
      const int r = i / COLS, c = i - r * COLS;
      const int y = yin0 + r, x = xin0 + c;
      const bool inside = (i < total) && y >= 0 && y < a.H && x >= 0 && x < a.W;
      const size_t off = inside ? (size_t)y * a.W + x : 0;
      dst[u] = (i < total) ? r * PITCH + c : -1;
      v[u] = inside ? a.in[off] : 0.f;
      sc[u] = (inside && a.in_scale) ? a.in_scale[off] : 1.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (dst[u] >= 0) win[dst[u]] = v[u] * sc[u];
  }
  __syncthreads();

  const int n = lane & 15, kk = lane >> 4;
  f32x4 acc[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const float* af = a.afrag + lane;
  float a_cur[STEPS], a_nxt[STEPS];
#pragma unroll
  for (int s = 0; s < STEPS; ++s) a_cur[s] = af[s * 64];

  // window row of output row (16 b + n) for PSF row dy is 16 b + n + (kh - 1 - dy)
  const float* bp = win + (n + a.kh - 1) * PITCH + wave * 16 + kk;
  for (int dy = 0; dy < a.kh; ++dy) {
    {  // unconditional prefetch of the next PSF row (the last iteration re-reads its own row): a
       // branch here would make the compiler drain every outstanding load (vmcnt(0))
      const int dn = dy + 1 < a.kh ? dy + 1 : dy;
#pragma unroll
      for (int s = 0; s < STEPS; ++s) a_nxt[s] = af[(dn * STEPS + s) * 64];
    }
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const float bv = bp[b * 16 * PITCH + 4 * s];
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[s], bv, acc[b], 0, 0, 0);
      }
    }
#pragma unroll
    for (int s = 0; s < STEPS; ++s) a_cur[s] = a_nxt[s];
    bp -= PITCH;
  }

  // ---- epilogue: lane holds out[y0 + 16 b + n][x0 + 16 wave + 4 kk + 0..3] -----------------------
  const int x = x0 + wave * 16 + 4 * kk;
  const bool vec = (a.W % 4 == 0) && (x + 3 < a.W);
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int y = y0 + 16 * b + n;
    if (y >= a.H || x >= a.W) continue;
    const size_t off = (size_t)y * a.W + x;
    float v[4] = {acc[b][0] * a.coef, acc[b][1] * a.coef, acc[b][2] * a.coef, acc[b][3] * a.coef};
    if (vec) {
      if (a.out_scale) {
        const float4 s4 = *reinterpret_cast<const float4*>(a.out_scale + off);
        v[0] *= s4.x, v[1] *= s4.y, v[2] *= s4.z, v[3] *= s4.w;
      }
      if (a.accumulate) {
        const float4 o4 = *reinterpret_cast<const float4*>(a.out + off);
        v[0] += o4.x, v[1] += o4.y, v[2] += o4.z, v[3] += o4.w;
      }
      *reinterpret_cast<float4*>(a.out + off) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (x + i >= a.W) break;
        float r = v[i];
        if (a.out_scale) r *= a.out_scale[off + i];
        if (a.accumulate) r += a.out[off + i];
        a.out[off + i] = r;
      }
    }
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

int direct_conv_kc(int kw) { return ((16 + kw - 1) + 3) / 4 * 4; }

bool direct_conv_supported(int kh, int kw) { return kh >= 1 && kw >= 1 && kh <= 33 && direct_conv_kc(kw) <= 48; }

size_t direct_conv_fragment_floats(int kh, int kw) { return (size_t)kh * (direct_conv_kc(kw) / 4) * 64; }

int launch_toeplitz_fragments(const float* psf, float* afrag_fwd, float* afrag_adj, int kh, int kw, hipStream_t stream) {
  const int steps = direct_conv_kc(kw) / 4;
  const int n = kh * steps * 64;
  toeplitz_fragments_kernel<<<(n + 255) / 256, 256, 0, stream>>>(psf, afrag_fwd, kh, kw, steps, 0);
  toeplitz_fragments_kernel<<<(n + 255) / 256, 256, 0, stream>>>(psf, afrag_adj, kh, kw, steps, 1);
  JD_LAUNCH_CHECK();
  return JD_OK;
}

template <int KC>
static int launch_kc(const DirectConvArgs& a, hipStream_t stream) {
  constexpr int PITCH = 48 + KC + 2;
  const size_t lds = (size_t)(TILE - 1 + a.kh) * PITCH * sizeof(float);
  dim3 grid((a.W + TILE - 1) / TILE, (a.H + TILE - 1) / TILE);
  direct_conv_kernel<KC><<<grid, 256, lds, stream>>>(a);
  JD_LAUNCH_CHECK();
  return JD_OK;
}

// adjoint == 0: out (+)= coef * out_scale * conv_same(in * in_scale, psf)    [crop offset (oy, ox)]
// adjoint != 0: out (+)= coef * out_scale * corr_same(in * in_scale, psf)    (the transpose of the above)
int launch_direct_conv(const float* in, const float* in_scale, const float* afrag, float* out, const float* out_scale,
                       int H, int W, int kh, int kw, int oy, int ox, int adjoint, float coef, int accumulate,
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
  switch (direct_conv_kc(kw)) {
    case 16: return launch_kc<16>(a, stream);
    case 20: return launch_kc<20>(a, stream);
    case 24: return launch_kc<24>(a, stream);
    case 28: return launch_kc<28>(a, stream);
    case 32: return launch_kc<32>(a, stream);
    case 36: return launch_kc<36>(a, stream);
    case 40: return launch_kc<40>(a, stream);
    case 44: return launch_kc<44>(a, stream);
    case 48: return launch_kc<48>(a, stream);
    default: return fail(JD_ERR_INVALID, "direct convolution: PSF width %d not supported", kw);
  }
}

}  // namespace jd
