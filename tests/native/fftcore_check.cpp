// CPU check of csrc/jd_fftcore.h: the Stockham passes and in-register butterflies of the native FFT convolution, run
// butterfly by butterfly on the host, against a float64 DFT.  Build + run: tests/test_fftcore_host.py.
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "jd_fftcore.h"

using namespace jdfft;

template <int DIR>
static void run_pass(int R, const float2* x, float2* y, int N, int p, const float2* tw) {
  for (int b = 0; b < N / R; ++b) switch (R) {
      case 16: pass_one<16, DIR>(x, y, N, p, tw, b); break;
      case 8: pass_one<8, DIR>(x, y, N, p, tw, b); break;
      case 4: pass_one<4, DIR>(x, y, N, p, tw, b); break;
      case 2: pass_one<2, DIR>(x, y, N, p, tw, b); break;
      case 9: pass_one<9, DIR>(x, y, N, p, tw, b); break;
      case 3: pass_one<3, DIR>(x, y, N, p, tw, b); break;
      default: std::abort();
    }
}

static double check(int N, int dir) {
  const Radices f = factorize(N);
  if (!f.n) return -1.0;
  std::vector<float2> tw(N), a(lp_size(N)), b(lp_size(N));
  for (int m = 0; m < N; ++m) tw[m] = float2{(float)std::cos(-2.0 * M_PI * m / N), (float)std::sin(-2.0 * M_PI * m / N)};
  std::vector<std::complex<double>> in(N), ref(N);
  srand(N + dir);
  for (int i = 0; i < N; ++i) {
    in[i] = {rand() / (double)RAND_MAX - 0.3, rand() / (double)RAND_MAX - 0.6};
    a[lp(i)] = float2{(float)in[i].real(), (float)in[i].imag()};
    in[i] = {(double)a[lp(i)].x, (double)a[lp(i)].y};
  }
  float2 *x = a.data(), *y = b.data();
  int p = 1;
  for (int s = 0; s < f.n; ++s) {
    if (dir < 0) run_pass<-1>(f.r[s], x, y, N, p, tw.data());
    else run_pass<1>(f.r[s], x, y, N, p, tw.data());
    p *= f.r[s];
    std::swap(x, y);
  }
  double norm = 0.0, err = 0.0;
  for (int k = 0; k < N; ++k) {
    std::complex<double> s = 0.0;
    for (int n = 0; n < N; ++n) s += in[n] * std::polar(1.0, dir * 2.0 * M_PI * ((long)k * n % N) / N);
    ref[k] = s;
    norm = std::fmax(norm, std::abs(s));
  }
  for (int k = 0; k < N; ++k) err = std::fmax(err, std::abs(std::complex<double>(x[lp(k)].x, x[lp(k)].y) - ref[k]));
  return err / norm;
}

int main() {
  int bad = 0;
  for (int N : {32, 48, 64, 96, 128, 144, 256, 288, 512, 576, 768, 1024, 1152, 2048, 2304, 4096, 4608}) {
    for (int dir : {-1, 1}) {
      const double e = check(N, dir);
      const Radices f = factorize(N);
      std::printf("N %5d dir %+d passes", N, dir);
      for (int s = 0; s < f.n; ++s) std::printf(" %d", f.r[s]);
      std::printf("  rel err %.2e%s\n", e, e >= 0 && e < 1e-6 ? "" : "  BAD");
      bad += !(e >= 0 && e < 1e-6);
    }
  }
  std::printf("next_length: 2064 -> %d, 1056 -> %d, 100 -> %d, 4200 -> %d\n", next_length(2064), next_length(1056), next_length(100), next_length(4200));
  return bad ? 1 : 0;
}
