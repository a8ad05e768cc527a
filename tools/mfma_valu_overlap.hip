// Micro-benchmark: does fp32 VALU work issued BETWEEN v_mfma_f32_16x16x4_f32 instructions of the same
// wave overlap with the matrix pipe, or does it add to the MFMA time?  One wave per SIMD (4 per CU),
// NV independent v_fma_f32 after every MFMA, 4 independent MFMA accumulator chains; then the same with two
// waves per SIMD (does one wave's VALU work hide behind the other wave's MFMAs?).
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/mfma_valu_overlap.hip -o /tmp/ovl && /tmp/ovl
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;

using f32x2 = __attribute__((ext_vector_type(2))) float;

// the same with NV packed v_pk_fma_f32 (2 fmas each) after every MFMA
template <int NV>
__global__ __launch_bounds__(512, 1) void kpk(float* out, int iters) {
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f + 1.f;
  f32x2 v[8];
  for (int i = 0; i < 8; ++i) v[i] = f32x2{a + i, b + i};
  const f32x2 m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u & 3], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) v[(u + j) & 7] = __builtin_elementwise_fma(v[(u + j) & 7], m, c);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int NV>
void runpk(float* out) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  kpk<NV><<<256, 256>>>(out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kpk<NV><<<256, 256>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("packed: NV=%d v_pk_fma_f32  %.3f ms  -> %.1f cycles per MFMA (+%d pk) at 2.4 GHz\n", NV, ms,
         ms * 1e-3 * 2.4e9 / (iters * 16.0), NV);
}

template <int NV>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters) {
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f + 1.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u & 3], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) v[(u + j) & 7] = fmaf(v[(u + j) & 7], 1.0001f, 0.5f);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int NV>
void run(float* out, int threads) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  k<NV><<<256, threads>>>(out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NV><<<256, threads>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma = (double)iters * 16;  // per wave
  const double cyc = ms * 1e-3 * 2.4e9 / mfma;
  printf("waves/SIMD=%d NV=%d  %.3f ms  -> %.1f cycles per MFMA of ONE wave (+%d v_fma) at 2.4 GHz\n", threads / 256, NV, ms, cyc, NV);
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * 4);
  for (int threads : {256, 512}) {
    run<0>(out, threads); run<2>(out, threads); run<4>(out, threads); run<8>(out, threads);
  }
  runpk<1>(out); runpk<2>(out); runpk<4>(out);
  return 0;
}
