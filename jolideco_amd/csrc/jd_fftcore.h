// Complex FFT of one sequence held in LDS, computed by the threads of one block: the building block of the native FFT
// convolution (fftnative.hip).  Stockham autosort passes (out of place, ping-pong between two buffers), one radix-R
// butterfly per thread and pass held in registers; radices 16 / 8 / 4 / 2 first, then one odd radix (9 or 3) last, so
// that the sub-transform length p of every pass is a power of two (k = b & (p - 1), no integer division).
// Lengths N = 2^a * {1, 3, 9}: 2304 = 16 * 16 * 9 covers a 2048-pixel axis plus PSF halos up to 513 taps at 12.5 % padding.
//
// The header compiles for the host too (JD_FFT_HD): tests/fftcore_check.cpp runs the same code thread by thread on
// the CPU against a float64 DFT.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define JD_FFT_HD __host__ __device__ __forceinline__
#else
#define JD_FFT_HD inline
struct float2 {
  float x, y;
};
#endif

namespace jdfft {

constexpr int MAX_PASSES = 6;

// padded index of element e of a sequence in LDS: one float2 of padding per 16 (the first pass writes R consecutive
// outputs per thread: a lane stride of 16 float2 would put every lane of a wave on the same banks)
JD_FFT_HD int lp(int e) { return e + (e >> 4); }
JD_FFT_HD int lp_size(int n) { return n + (n >> 4) + 1; }

JD_FFT_HD float2 cadd(float2 a, float2 b) { return float2{a.x + b.x, a.y + b.y}; }
JD_FFT_HD float2 csub(float2 a, float2 b) { return float2{a.x - b.x, a.y - b.y}; }
JD_FFT_HD float2 cmul(float2 a, float2 b) { return float2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
JD_FFT_HD float2 cconj(float2 a) { return float2{a.x, -a.y}; }
// multiplication by -i (DIR < 0: the forward transform's quarter turn) or +i (DIR > 0: the inverse's)
template <int DIR>
JD_FFT_HD float2 crot(float2 a) {
  return DIR < 0 ? float2{a.y, -a.x} : float2{-a.y, a.x};
}

// in-register DFTs, natural order in and out; DIR = -1: exp(-2 pi i jk / R) (forward), +1: inverse (unnormalised)
template <int R, int DIR>
struct Dft;

template <int DIR>
struct Dft<2, DIR> {
  static JD_FFT_HD void run(float2* u) {
    const float2 a = u[0], b = u[1];
    u[0] = cadd(a, b), u[1] = csub(a, b);
  }
};

template <int DIR>
struct Dft<4, DIR> {
  static JD_FFT_HD void run(float2* u) {
    const float2 t0 = cadd(u[0], u[2]), t1 = csub(u[0], u[2]), t2 = cadd(u[1], u[3]), t3 = crot<DIR>(csub(u[1], u[3]));
    u[0] = cadd(t0, t2), u[1] = cadd(t1, t3), u[2] = csub(t0, t2), u[3] = csub(t1, t3);
  }
};

// radix-2 decimation in time on top of two half-size transforms (R = 8, 16)
template <int R, int DIR>
struct Dft {
  static JD_FFT_HD void run(float2* u) {
    constexpr int H = R / 2;
    float2 e[H], o[H];
#pragma unroll
    for (int i = 0; i < H; ++i) e[i] = u[2 * i], o[i] = u[2 * i + 1];
    Dft<H, DIR>::run(e);
    Dft<H, DIR>::run(o);
#pragma unroll
    for (int k = 0; k < H; ++k) {
      // w = exp(DIR * 2 pi i k / R) (compile-time constants after unrolling)
      const double ang = (DIR < 0 ? -1.0 : 1.0) * 6.283185307179586476925286766559 * k / R;
      float2 t;
      if (k == 0) t = o[k];
      else if (4 * k == R) t = crot<DIR>(o[k]);
      else t = cmul(o[k], float2{(float)__builtin_cos(ang), (float)__builtin_sin(ang)});
      u[k] = cadd(e[k], t), u[k + H] = csub(e[k], t);
    }
  }
};

template <int DIR>
struct Dft<3, DIR> {
  static JD_FFT_HD void run(float2* u) {
    const float c = 0.86602540378443864676f;  // sin(2 pi / 3)
    const float2 s = cadd(u[1], u[2]), d = csub(u[1], u[2]);
    const float2 m = float2{u[0].x - 0.5f * s.x, u[0].y - 0.5f * s.y};
    const float2 r = crot<DIR>(float2{c * d.x, c * d.y});
    u[0] = cadd(u[0], s), u[1] = cadd(m, r), u[2] = csub(m, r);
  }
};

template <int DIR>
struct Dft<9, DIR> {
  static JD_FFT_HD void run(float2* u) {
    // n = 3 n1 + n2, k = k1 + 3 k2:  X[k1 + 3 k2] = sum_n2 w9^(n2 k1) w3^(n2 k2) sum_n1 w3^(n1 k1) x[3 n1 + n2]
    float2 a[3][3];
#pragma unroll
    for (int n2 = 0; n2 < 3; ++n2) {
      float2 v[3] = {u[n2], u[3 + n2], u[6 + n2]};
      Dft<3, DIR>::run(v);
#pragma unroll
      for (int k1 = 0; k1 < 3; ++k1) {
        if (n2 * k1 == 0) {
          a[n2][k1] = v[k1];
        } else {
          const double ang = (DIR < 0 ? -1.0 : 1.0) * 6.283185307179586476925286766559 * (n2 * k1) / 9.0;
          a[n2][k1] = cmul(v[k1], float2{(float)__builtin_cos(ang), (float)__builtin_sin(ang)});
        }
      }
    }
#pragma unroll
    for (int k1 = 0; k1 < 3; ++k1) {
      float2 v[3] = {a[0][k1], a[1][k1], a[2][k1]};
      Dft<3, DIR>::run(v);
#pragma unroll
      for (int k2 = 0; k2 < 3; ++k2) u[k1 + 3 * k2] = v[k2];
    }
  }
};

// One Stockham pass of radix R over a sequence of N elements: x -> y (both in padded LDS layout), sub-transform length
// p (a power of two; the product of the radices of the passes before).  `tw`: exp(-2 pi i m / N), m < N.
// Butterfly b of N / R: inputs x[b + t N / R], twiddles w^(t k) with k = b mod p, w = exp(DIR 2 pi i / (p R)), outputs
// y[(b - k) R + k + t p].
// LIN: the padded index is LINEAR in t on both sides -- inputs when N / R is a multiple of 16 (lp(b + t nb) = lp(b) +
// t (nb + nb / 16)), outputs when p = 1 (the R <= 16 outputs of a butterfly share one group of 16) or p is a multiple
// of 16 -- one base address each plus compile-time multiples of a uniform step instead of an add, a shift and an add per
// element.
JD_FFT_HD bool pass_is_linear(int N, int R, int p) { return ((N / R) & 15) == 0 && (p == 1 || (p & 15) == 0); }

template <int R, int DIR, bool LIN = false>
JD_FFT_HD void pass_one(const float2* x, float2* y, int N, int p, const float2* tw, int b) {
  const int nb = N / R;
  const int k = b & (p - 1);
  float2 u[R];
  if (LIN) {
    const float2* xb = x + lp(b);
    const int step = nb + (nb >> 4);
#pragma unroll
    for (int t = 0; t < R; ++t) u[t] = xb[t * step];
  } else {
#pragma unroll
    for (int t = 0; t < R; ++t) u[t] = x[lp(b + t * nb)];
  }
  if (p > 1) {
    // w^t built from one table entry by a product tree (depth <= 4: a few ulp), not by R - 1 dependent products
    float2 w[R];
    w[1] = tw[k * (nb / p)];
    if (DIR > 0) w[1].y = -w[1].y;
#pragma unroll
    for (int t = 2; t < R; ++t) w[t] = cmul(w[t / 2], w[t - t / 2]);
#pragma unroll
    for (int t = 1; t < R; ++t) u[t] = cmul(u[t], w[t]);
  }
  Dft<R, DIR>::run(u);
  const int j = (b - k) * R + k;
  if (LIN) {
    float2* yb = y + lp(j);
    const int step = p == 1 ? 1 : p + (p >> 4);
#pragma unroll
    for (int t = 0; t < R; ++t) yb[t * step] = u[t];
  } else {
#pragma unroll
    for (int t = 0; t < R; ++t) y[lp(j + t * p)] = u[t];
  }
}

struct Radices {
  int n, r[MAX_PASSES];
};

// radix schedule of N = 2^a * {1, 3, 9}, a >= 3: radices 16 and 8 (one 4 where a = 5 leaves no other way), the odd
// radix last; n = 0: not supported
inline Radices factorize(int N) {
  Radices f{};
  int m = N, odd = 1;
  while (m % 3 == 0 && odd < 9) m /= 3, odd *= 3;
  if (m < 8 || (m & (m - 1)) != 0) return Radices{};
  int a = 0;
  while ((1 << a) < m) ++a;
  // a = 4 i + 3 j (+ 2 for one radix 4 when a = 5): as many 16s as possible
  int n16 = a / 4, rest = a - 4 * n16, n8 = 0, n4 = 0;
  while (rest % 3 != 0 && n16 > 0) --n16, rest += 4;
  if (rest % 3 == 0) n8 = rest / 3;
  else if (a == 5) n16 = 0, n8 = 1, n4 = 1;
  else return Radices{};
  if (n16 + n8 + n4 + (odd > 1) > MAX_PASSES) return Radices{};
  for (int i = 0; i < n16; ++i) f.r[f.n++] = 16;
  for (int i = 0; i < n8; ++i) f.r[f.n++] = 8;
  for (int i = 0; i < n4; ++i) f.r[f.n++] = 4;
  if (odd > 1) f.r[f.n++] = odd;
  return f;
}

// smallest supported length >= n; with_three: lengths 3 * 2^a too (the row transforms; the column transforms, one wave
// per sequence, keep to 2^a * {1, 9})
inline int next_length(int n, bool with_three = true) {
  int best = 0;
  for (int odd : {1, 3, 9}) {
    if (odd == 3 && !with_three) continue;
    for (int m = 8; m <= (1 << 20); m *= 2) {
      const long v = (long)m * odd;
      if (v >= n && v >= 32 && factorize((int)v).n > 0) {
        if (!best || v < best) best = (int)v;
        break;
      }
    }
  }
  return best;
}

}  // namespace jdfft
