// In-library kernel timers: hipEvent pairs recorded around the named kernels ON THE STREAM THE
// KERNEL IS LAUNCHED ON, so a benchmark can report the average launch duration of a kernel inside
// its timed region without a profiler attached (bench.py `roofline`).  Disabled by default: the
// record calls cost nothing unless jd_profile_enable() was called.
#include <algorithm>
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

namespace jd {
// Sustained shader clock: every CU runs a dependent fp32 FMA chain for `iters` rounds; one lane per block stamps the shader
// clock (s_memtime: one tick per shader cycle) and the constant 100 MHz reference clock (s_memrealtime) around it
// (MI355X_MICROARCH.md, DVFS give-back item 6): MHz = 100 * d(memtime) / d(memrealtime).
__global__ __launch_bounds__(256) void clock_probe_kernel(unsigned long long* stamps, float* sink, int iters) {
  float a = 1.0f + 1e-7f * threadIdx.x, b = 0.5f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 64; ++k) b = fmaf(a, b, 1e-9f);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t1 - t0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
  if (b == 123.456f) sink[0] = b;  // keeps the chain alive
}
}  // namespace jd

extern "C" int jd_clock_probe(double milliseconds, double* mhz_out, void* stream) {
  JD_REQUIRE(mhz_out && milliseconds > 0.0 && milliseconds <= 100.0, "jd_clock_probe: bad argument");
  hipStream_t s = as_stream(stream);
  int dev = 0;
  hipDeviceProp_t prop;
  JD_HIP(hipGetDevice(&dev));
  JD_HIP(hipGetDeviceProperties(&prop, dev));
  const int blocks = prop.multiProcessorCount * 4;  // four blocks of four waves per CU: every SIMD has four waves
  unsigned long long* stamps = nullptr;
  float* sink = nullptr;
  JD_HIP(hipMalloc(&stamps, (size_t)blocks * 2 * sizeof(unsigned long long)));
  JD_HIP(hipMalloc(&sink, sizeof(float)));
  // 64 dependent FMAs per round at 4 waves per SIMD: ~8 cycles per FMA and wave -> ~0.25 us per round at 2 GHz
  const int iters = (int)(milliseconds * 1e3 / 0.25);
  clock_probe_kernel<<<blocks, 256, 0, s>>>(stamps, sink, iters > 1 ? iters : 1);
  hipError_t e = hipGetLastError();
  std::vector<unsigned long long> host((size_t)blocks * 2);
  if (e == hipSuccess) e = hipMemcpyAsync(host.data(), stamps, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hipFree(stamps);
  (void)hipFree(sink);
  if (e != hipSuccess) return fail(JD_ERR_HIP, "jd_clock_probe: %s", hipGetErrorString(e));
  std::vector<double> mhz;
  for (int b = 0; b < blocks; ++b)
    if (host[2 * b + 1]) mhz.push_back(100.0 * (double)host[2 * b] / (double)host[2 * b + 1]);
  JD_REQUIRE(!mhz.empty(), "jd_clock_probe: no stamps");
  std::sort(mhz.begin(), mhz.end());
  *mhz_out = mhz[mhz.size() / 2];  // median over the blocks
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
    case JD_KERNEL_SHIFT: return "shift_kernels";
    default: return "?";
  }
}
