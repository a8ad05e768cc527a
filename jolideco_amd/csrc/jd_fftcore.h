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

// Complex arithmetic.  On the device a complex number is a PAIR of 32-bit registers and every operation below is one or
// two PACKED fp32 instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 work on both halves of a register pair; their
// op_sel / op_sel_hi modifiers pick, per result half, which half of each source is read, neg_lo / neg_hi negate a source
// per result half): an addition is ONE instruction instead of two, a complex product TWO instead of four, and a
// multiplication by +-i costs nothing (it is the operand swizzle of the addition that consumes it).  The transforms of
// this library are bound by their vector-instruction count (fftnative.hip), so this halves their arithmetic.  On the
// host (tests/native/fftcore_check.cpp) the same functions are plain scalar code.
#if defined(__HIP_DEVICE_COMPILE__)
#define JD_FFT_PACKED 1
typedef float cf __attribute__((ext_vector_type(2)));
#else
#define JD_FFT_PACKED 0
typedef float2 cf;
#endif

JD_FFT_HD cf cadd(cf a, cf b) {
#if JD_FFT_PACKED
  return a + b;
#else
  return cf{a.x + b.x, a.y + b.y};
#endif
}
JD_FFT_HD cf csub(cf a, cf b) {
#if JD_FFT_PACKED
  return a - b;
#else
  return cf{a.x - b.x, a.y - b.y};
#endif
}
// a * b
JD_FFT_HD cf cmul(cf a, cf b) {
#if JD_FFT_PACKED
  cf t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));  // (a.x b.x, a.x b.y)
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));  // (-a.y b.y, a.y b.x) + t
  return r;
#else
  return cf{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
#endif
}
// a * (c + i s) for a compile-time constant: the constant lives in a scalar register pair
JD_FFT_HD cf cmul_const(cf a, float c, float s) {
#if JD_FFT_PACKED
  const cf b = cf{c, s};
  cf t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "s"(b));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "s"(b), "v"(t));
  return r;
#else
  return cf{a.x * c - a.y * s, a.x * s + a.y * c};
#endif
}
// a + c * b, c real
JD_FFT_HD cf cfma_real(cf a, float c, cf b) {
#if JD_FFT_PACKED
  return a + b * cf{c, c};
#else
  return cf{a.x + c * b.x, a.y + c * b.y};
#endif
}
JD_FFT_HD cf cscale(cf a, float c) {
#if JD_FFT_PACKED
  return a * cf{c, c};
#else
  return cf{a.x * c, a.y * c};
#endif
}
JD_FFT_HD cf cconj(cf a) { return cf{a.x, -a.y}; }
// multiplication by -i (DIR < 0: the forward transform's quarter turn) or +i (DIR > 0: the inverse's)
template <int DIR>
JD_FFT_HD cf crot(cf a) {
  return DIR < 0 ? cf{a.y, -a.x} : cf{-a.y, a.x};
}
// a + crot<DIR>(b) and a - crot<DIR>(b): the quarter turn is the operand swizzle of the addition
template <int DIR>
JD_FFT_HD cf cadd_rot(cf a, cf b) {
#if JD_FFT_PACKED
  cf r;
  if (DIR < 0) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));  // (a.x + b.y, a.y - b.x)
  else asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));         // (a.x - b.y, a.y + b.x)
  return r;
#else
  return cadd(a, crot<DIR>(b));
#endif
}
template <int DIR>
JD_FFT_HD cf csub_rot(cf a, cf b) {
  return cadd_rot<-DIR>(a, b);
}

// in-register DFTs, natural order in and out; DIR = -1: exp(-2 pi i jk / R) (forward), +1: inverse (unnormalised)
template <int R, int DIR>
struct Dft;

template <int DIR>
struct Dft<2, DIR> {
  static JD_FFT_HD void run(cf* u) {
    const cf a = u[0], b = u[1];
    u[0] = cadd(a, b), u[1] = csub(a, b);
  }
};

template <int DIR>
struct Dft<4, DIR> {
  static JD_FFT_HD void run(cf* u) {
    const cf t0 = cadd(u[0], u[2]), t1 = csub(u[0], u[2]), t2 = cadd(u[1], u[3]), d = csub(u[1], u[3]);
    u[0] = cadd(t0, t2), u[1] = cadd_rot<DIR>(t1, d), u[2] = csub(t0, t2), u[3] = csub_rot<DIR>(t1, d);
  }
};

// radix-2 decimation in time on top of two half-size transforms (R = 8, 16)
template <int R, int DIR>
struct Dft {
  static JD_FFT_HD void run(cf* u) {
    constexpr int H = R / 2;
    cf e[H], o[H];
#pragma unroll
    for (int i = 0; i < H; ++i) e[i] = u[2 * i], o[i] = u[2 * i + 1];
    Dft<H, DIR>::run(e);
    Dft<H, DIR>::run(o);
#pragma unroll
    for (int k = 0; k < H; ++k) {
      // w = exp(DIR * 2 pi i k / R) (compile-time constants after unrolling)
      const double ang = (DIR < 0 ? -1.0 : 1.0) * 6.283185307179586476925286766559 * k / R;
      if (k == 0) {
        u[k] = cadd(e[k], o[k]), u[k + H] = csub(e[k], o[k]);
      } else if (4 * k == R) {
        u[k] = cadd_rot<DIR>(e[k], o[k]), u[k + H] = csub_rot<DIR>(e[k], o[k]);
      } else {
        const cf t = cmul_const(o[k], (float)__builtin_cos(ang), (float)__builtin_sin(ang));
        u[k] = cadd(e[k], t), u[k + H] = csub(e[k], t);
      }
    }
  }
};

template <int DIR>
struct Dft<3, DIR> {
  static JD_FFT_HD void run(cf* u) {
    const float c = 0.86602540378443864676f;  // sin(2 pi / 3)
    const cf s = cadd(u[1], u[2]), d = cscale(csub(u[1], u[2]), c);
    const cf m = cfma_real(u[0], -0.5f, s);
    u[0] = cadd(u[0], s), u[1] = cadd_rot<DIR>(m, d), u[2] = csub_rot<DIR>(m, d);
  }
};

template <int DIR>
struct Dft<9, DIR> {
  static JD_FFT_HD void run(cf* u) {
    // n = 3 n1 + n2, k = k1 + 3 k2:  X[k1 + 3 k2] = sum_n2 w9^(n2 k1) w3^(n2 k2) sum_n1 w3^(n1 k1) x[3 n1 + n2]
    cf a[3][3];
#pragma unroll
    for (int n2 = 0; n2 < 3; ++n2) {
      cf v[3] = {u[n2], u[3 + n2], u[6 + n2]};
      Dft<3, DIR>::run(v);
#pragma unroll
      for (int k1 = 0; k1 < 3; ++k1) {
        if (n2 * k1 == 0) {
          a[n2][k1] = v[k1];
        } else {
          const double ang = (DIR < 0 ? -1.0 : 1.0) * 6.283185307179586476925286766559 * (n2 * k1) / 9.0;
          a[n2][k1] = cmul_const(v[k1], (float)__builtin_cos(ang), (float)__builtin_sin(ang));
        }
      }
    }
#pragma unroll
    for (int k1 = 0; k1 < 3; ++k1) {
      cf v[3] = {a[0][k1], a[1][k1], a[2][k1]};
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
JD_FFT_HD constexpr bool pass_is_linear(int N, int R, int p) { return ((N / R) & 15) == 0 && (p == 1 || (p & 15) == 0); }

template <int R, int DIR, bool LIN = false>
JD_FFT_HD void pass_one(const float2* x_, float2* y_, int N, int p, const float2* tw_, int b) {
  const cf* x = reinterpret_cast<const cf*>(x_);
  cf* y = reinterpret_cast<cf*>(y_);
  const cf* tw = reinterpret_cast<const cf*>(tw_);
  const int nb = N / R;
  const int k = b & (p - 1);
  cf u[R];
  if (LIN) {
    const cf* xb = x + lp(b);
    const int step = nb + (nb >> 4);
#pragma unroll
    for (int t = 0; t < R; ++t) u[t] = xb[t * step];
  } else {
#pragma unroll
    for (int t = 0; t < R; ++t) u[t] = x[lp(b + t * nb)];
  }
  if (p > 1) {
    // w^t built from one table entry by a product tree (depth <= 4: a few ulp), not by R - 1 dependent products
    cf w[R];
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
    cf* yb = y + lp(j);
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
