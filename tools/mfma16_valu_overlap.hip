// Micro-benchmark: how many VALU instructions hide in the shadow of v_mfma_f32_32x32x16_f16 (the screen kernel's
// matrix instruction, 32 cycles/SIMD) when both come from the SAME wave, one wave per SIMD?  NV independent VALU
// instructions are placed after every MFMA (inline asm: the order in the binary is the order written here).
// Variants: v_fma_f32 on plain registers, v_fma_f32 squaring the registers of a finished accumulator (what the
// screen's epilogue does), v_pk_fma_f32.  Reports cycles per MFMA from s_memtime (shader clock) and from the wall
// clock (the ratio is the effective clock under this load).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma16_valu_overlap.hip -o /tmp/ovl16 && /tmp/ovl16
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

#define MFMA(acc, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

// MODE 0: fma on private registers; 1: squares of the other accumulator's registers (read-only) into 2 chains;
// 2: v_pk_fma_f32 on private registers
template <int NV, int MODE>
__global__ __launch_bounds__(512, 1) void kern(float* out, long long* cyc, int iters, const _Float16* src) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) a[e] = src[(threadIdx.x * 8 + e) & 4095], b[e] = src[(threadIdx.x * 8 + e + 77) & 4095];
  float v[8];
  f32x2 p[4];
  for (int i = 0; i < 8; ++i) v[i] = 1e-3f * threadIdx.x + i;
  for (int i = 0; i < 4; ++i) p[i] = f32x2{v[i], v[i + 4]};
  const float m = 1.0001f, c = 0.5f;
  const f32x2 m2 = {1.0001f, 0.9999f}, c2 = {0.5f, 0.25f};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      MFMA(acc[u & 3], a, b);
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int idx = (u * NV + j);
        if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[idx & 7]) : "v"(m), "v"(c));
        if (MODE == 1) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(v[idx & 1]) : "v"(acc[(u + 2) & 3][idx & 15]));
        if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[idx & 3]) : "v"(m2), "v"(c2));
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int i = 0; i < 4; ++i) s += p[i][0] + p[i][1];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NV, int MODE>
void run(float* out, long long* cyc, const _Float16* src, int threads = 256) {
  const int iters = 20000, blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  kern<NV, MODE><<<blocks, threads>>>(out, cyc, 100, src);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<NV, MODE><<<blocks, threads>>>(out, cyc, iters, src);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks * (threads / 64));
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto x : h) mean += (double)x;
  mean /= h.size();
  const double n_mfma = (double)iters * 12;
  const char* names[3] = {"v_fma_f32 (private regs)", "v_fma_f32 (squares of a finished accumulator)", "v_pk_fma_f32"};
  // s_memtime ticks at 100 MHz on gfx9 (constant clock); shader cycles = wall time x shader clock, unknown a priori
  printf("waves/SIMD %d %-46s NV=%2d: %8.3f ms, %6.2f ns per MFMA slot, memtime ticks per MFMA %.3f\n", threads / 256, names[MODE], NV, ms,
         ms * 1e6 / n_mfma, mean / n_mfma);
}

int main() {
  float* out;
  long long* cyc;
  _Float16* src;
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&cyc, 256 * 8 * 8);
  hipMalloc(&src, 4096 * 2);
  std::vector<_Float16> h(4096);
  unsigned s = 12345u;
  for (auto& x : h) s = s * 1664525u + 1013904223u, x = (_Float16)(((s >> 8) & 0xFFFF) / 65536.0f - 0.5f);
  hipMemcpy(src, h.data(), 4096 * 2, hipMemcpyHostToDevice);
  run<0, 0>(out, cyc, src);
  run<2, 0>(out, cyc, src);
  run<4, 0>(out, cyc, src);
  run<5, 0>(out, cyc, src);
  run<6, 0>(out, cyc, src);
  run<8, 0>(out, cyc, src);
  run<12, 0>(out, cyc, src);
  run<4, 1>(out, cyc, src);
  run<6, 1>(out, cyc, src);
  run<8, 1>(out, cyc, src);
  run<2, 2>(out, cyc, src);
  run<4, 2>(out, cyc, src);
  // two waves per SIMD: do the VALU instructions of one wave issue in the shadow of the other's MFMAs?
  run<0, 0>(out, cyc, src, 512);
  run<4, 0>(out, cyc, src, 512);
  run<8, 0>(out, cyc, src, 512);
  run<12, 0>(out, cyc, src, 512);
  run<16, 0>(out, cyc, src, 512);
  run<8, 1>(out, cyc, src, 512);
  run<12, 1>(out, cyc, src, 512);
  return 0;
}
