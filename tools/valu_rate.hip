// Micro-benchmark: issue rate of v_fma_f32 against v_pk_fma_f32 (and v_pk_mul_f32) on gfx950, no MFMAs around: NI
// independent accumulators per lane, W waves per SIMD.  Prints ns per instruction per SIMD (wall clock) -- the ratio
// packed / scalar says whether a packed fp32 FMA costs one issue slot (full rate: twice the flops) or two.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>

#include <cstdio>
using f32x2 = __attribute__((ext_vector_type(2))) float;

template <int MODE>
__global__ __launch_bounds__(1024) void kern(float* out, int iters) {
  float v[8];
  f32x2 p[8];
  for (int i = 0; i < 8; ++i) v[i] = 1e-3f * threadIdx.x + i, p[i] = f32x2{v[i], v[i] + 0.5f};
  const float m = 1.0001f, c = 0.5f;
  const f32x2 m2 = {1.0001f, 0.9999f}, c2 = {0.5f, 0.25f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(m), "v"(c));
        if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));
        if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(m2));
        if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(m2), "v"(c2));  // accumulate form
      }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i] + p[i][0] + p[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(float* out, int threads, const char* name) {
  const int iters = 20000, blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  kern<MODE><<<blocks, threads>>>(out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<MODE><<<blocks, threads>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double n_inst = (double)iters * 32 * (threads / 256);  // per SIMD
  printf("%-28s waves/SIMD %d: %8.3f ms, %6.3f ns per instruction per SIMD\n", name, threads / 256, ms, ms * 1e6 / n_inst);
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 1024 * 4);
  for (int threads : {256, 512, 1024}) {
    run<0>(out, threads, "v_fma_f32");
    run<1>(out, threads, "v_pk_fma_f32 (x*m+c)");
    run<3>(out, threads, "v_pk_fma_f32 (m*c+acc)");
    run<2>(out, threads, "v_pk_mul_f32");
  }
  return 0;
}
