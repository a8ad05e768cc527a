// Internal launch prototypes shared between the translation units of libjolideco_hip.so.
#pragma once
#include "jd_common.h"

namespace jd {

struct PoissonArgs {
  const float* conv[JD_MAX_COMPONENTS];  // padded (Hp, Wp) convolution results, read at the crop offset
  float* g[JD_MAX_COMPONENTS];           // padded (Hp, Wp) outputs: masked d loss / d conv_c
  const float* background;
  const float* counts;
  float* npred_out;  // nullable
  double* partials;
  int n_comp, H, W, Hp, Wp, oy, ox;  // (H, W): the conv / flux grid; counts live on (H / up, W / up)
  float eps, inv_n;
  int write_grad;
  int up;  // up-sampling factor of the flux grid w.r.t. the counts grid (1 = none)
};

int launch_pad_mul(const float* image, const float* scale, float* padded, int H, int W, int Hp, int Wp,
                   hipStream_t stream);
int launch_cmul(float2* spec, const float2* khat, size_t n, bool conj, hipStream_t stream);
int launch_poisson_fused(const PoissonArgs& a, int* n_partials, hipStream_t stream);
int launch_poisson_pooled(const PoissonArgs& a, int* n_partials, hipStream_t stream);
int poisson_fused_max_partials(int Hp, int Wp);
int launch_adjoint_epilogue(const float* corr, const float* scale, float* grad, int H, int W, int Hp,
                            int Wp, int oy, int ox, float coef, int accumulate, hipStream_t stream);
int launch_crop(const float* padded, float* out, int H, int W, int Wp, int oy, int ox, hipStream_t stream);

// direct (MFMA Toeplitz) convolution for small PSFs (directconv.hip)
enum { JD_CONV_FFT = 0, JD_CONV_DIRECT = 1 };
bool direct_conv_supported(int kh, int kw);
size_t direct_conv_fragment_floats(int kh, int kw);
int launch_toeplitz_fragments(const float* psf, float* afrag_fwd, float* afrag_adj, int kh, int kw, hipStream_t stream);
int launch_direct_conv(const float* in, const float* in_scale, const float* afrag, float* out, const float* out_scale,
                       int H, int W, int kh, int kw, int oy, int ox, int adjoint, float coef, int accumulate,
                       hipStream_t stream);

// kernel timers (profile.hip): RAII bracket around one launch
int prof_begin(int kernel, hipStream_t s);
void prof_end(int slot, hipStream_t s);
struct ProfScope {
  int slot;
  hipStream_t s;
  ProfScope(int kernel, hipStream_t stream) : slot(prof_begin(kernel, stream)), s(stream) {}
  ~ProfScope() { prof_end(slot, s); }
};

}  // namespace jd
