// In-library kernel timers: hipEvent pairs recorded around the named kernels ON THE STREAM THE
// KERNEL IS LAUNCHED ON, so a benchmark can report the average launch duration of a kernel inside
// its timed region without a profiler attached (bench.py `roofline`).  Disabled by default: the
// record calls cost nothing unless jd_profile_enable() was called.
#include <vector>

#include "jd_common.h"
#include "kernels.h"

namespace jd {

namespace {
struct Pair {
  hipEvent_t start = nullptr, stop = nullptr;
  int kernel = -1;
};
std::vector<Pair> g_pairs;  // pool, created by jd_profile_enable
size_t g_used = 0;
bool g_enabled = false;
size_t g_dropped = 0;
}  // namespace

int prof_begin(int kernel, hipStream_t s) {
  if (!g_enabled) return -1;
  if (g_used >= g_pairs.size()) {
    ++g_dropped;
    return -1;
  }
  Pair& p = g_pairs[g_used];
  p.kernel = kernel;
  if (hipEventRecord(p.start, s) != hipSuccess) return -1;
  return (int)g_used++;
}

void prof_end(int slot, hipStream_t s) {
  if (slot >= 0) (void)hipEventRecord(g_pairs[(size_t)slot].stop, s);
}

}  // namespace jd

using namespace jd;

extern "C" int jd_profile_enable(int capacity) {
  JD_REQUIRE(capacity > 0 && capacity <= (1 << 20), "jd_profile_enable: capacity %d out of range", capacity);
  while (g_pairs.size() < (size_t)capacity) {
    Pair p;
    // no system-scope fence / cache flush at the record: the timers should perturb the stream as
    // little as possible (a default event costs ~3 us per pair on a 12 us kernel)
    JD_HIP(hipEventCreateWithFlags(&p.start, hipEventDisableSystemFence));
    JD_HIP(hipEventCreateWithFlags(&p.stop, hipEventDisableSystemFence));
    g_pairs.push_back(p);
  }
  g_used = 0;
  g_dropped = 0;
  g_enabled = true;
  return JD_OK;
}

extern "C" int jd_profile_disable(void) {
  g_enabled = false;
  return JD_OK;
}

extern "C" int jd_profile_pause(int paused) {
  // keeps the recorded pairs; launches issued while paused are simply not timed (a benchmark can
  // sample every n-th step and keep the event overhead out of the others)
  g_enabled = !paused && !g_pairs.empty();
  return JD_OK;
}

extern "C" int jd_profile_read(int kernel, double* total_ms, long long* launches) {
  JD_REQUIRE(total_ms && launches, "jd_profile_read: null argument");
  JD_REQUIRE(kernel >= 0 && kernel < JD_KERNEL_COUNT, "jd_profile_read: unknown kernel id %d", kernel);
  double total = 0.0;
  long long n = 0;
  for (size_t i = 0; i < g_used; ++i) {
    if (g_pairs[i].kernel != kernel) continue;
    JD_HIP(hipEventSynchronize(g_pairs[i].stop));
    float ms = 0.f;
    JD_HIP(hipEventElapsedTime(&ms, g_pairs[i].start, g_pairs[i].stop));
    total += (double)ms;
    ++n;
  }
  *total_ms = total;
  *launches = n;
  return JD_OK;
}

extern "C" const char* jd_kernel_name(int kernel) {
  switch (kernel) {
    case JD_KERNEL_POISSON_FUSED: return "poisson_fused_kernel";
    case JD_KERNEL_GMM_FWD: return "gmm_fwd_kernel";
    case JD_KERNEL_GMM_BWD: return "gmm_bwd_max_kernel";
    case JD_KERNEL_GMM_GATHER: return "gmm_gather_kernel";
    case JD_KERNEL_PAD_MUL: return "pad_mul_kernel";
    case JD_KERNEL_CMUL: return "cmul_kernel";
    case JD_KERNEL_ADJOINT_EPILOGUE: return "adjoint_epilogue_kernel";
    case JD_KERNEL_ADAM: return "adam_kernel";
    case JD_KERNEL_FFT_R2C: return "rocfft_r2c";
    case JD_KERNEL_FFT_C2R: return "rocfft_c2r";
    case JD_KERNEL_DIRECT_CONV: return "direct_conv_kernel";
    case JD_KERNEL_SEP_CONV: return "sep_conv_kernel";
    case JD_KERNEL_GMM_SCREEN: return "gmm_screen_kernel";
    case JD_KERNEL_GMM_SORT: return "gmm_bucket_kernels";
    case JD_KERNEL_GMM_EXACT: return "gmm_exact_kernel";
    case JD_KERNEL_GMM_STAGE: return "gmm_stage_kernel";
    default: return "?";
  }
}
