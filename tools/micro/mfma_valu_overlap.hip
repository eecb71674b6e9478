// Does one wave's VALU / transcendental stream overlap its own MFMAs on gfx950?  Times (s_memtime) a loop of
//   mode 0: 12 x v_mfma_f32_32x32x16_bf16            mode 1: 32 x v_exp_f32 + 48 plain VALU
//   mode 2: both, MFMA first then VALU               mode 3: both, interleaved (1 MFMA : 4 exp + 2 VALU, fenced)
//   mode 4: 12 MFMA + 80 plain VALU interleaved      mode 5: 80 plain VALU only
// with 1 or 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define FENCE() __builtin_amdgcn_sched_barrier(0)

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(0.5f + i); }
  float v[32];
  for (int i = 0; i < 32; ++i) v[i] = threadIdx.x * 1e-3f + i * 1e-2f;
  float w[16];
  for (int i = 0; i < 16; ++i) w[i] = 1.0f + i;
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 2) {
#pragma unroll
      for (int m = 0; m < 12; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
    }
    if (MODE == 1 || MODE == 2) {
#pragma unroll
      for (int i = 0; i < 32; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]) * 0.999f;       // 32 exp + 32 mul
#pragma unroll
      for (int i = 0; i < 16; ++i) w[i] = w[i] * 1.0001f + 0.5f;                         // 16 fma
    }
    if (MODE == 3) {
#pragma unroll
      for (int m = 0; m < 12; ++m) {
        FENCE();
        acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
        FENCE();
        if (m < 8) {
#pragma unroll
          for (int i = 4 * m; i < 4 * m + 4; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]) * 0.999f;
#pragma unroll
          for (int i = 2 * m; i < 2 * m + 2; ++i) w[i] = w[i] * 1.0001f + 0.5f;
        }
      }
      FENCE();
    }
    if (MODE == 4) {
#pragma unroll
      for (int m = 0; m < 12; ++m) {
        FENCE();
        acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
        FENCE();
#pragma unroll
        for (int i = 0; i < 7; ++i) { const int j = (m * 7 + i) & 31; v[j] = v[j] * 1.0001f + 0.5f; }
      }
      FENCE();
    }
    if (MODE == 5) {
#pragma unroll
      for (int r = 0; r < 84; ++r) { const int j = r & 31; v[j] = v[j] * 1.0001f + 0.5f; }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i];
  for (int i = 0; i < 32; ++i) s += v[i];
  for (int i = 0; i < 16; ++i) s += w[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, iters); hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s waves/SIMD %d: %8.1f memtime ticks/iter (100 MHz), %7.3f us/iter-total %8.1f ns/iter\n", name, threads / 256, (double)c / iters, ms * 1e3, ms * 1e6 / iters);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int th : {256, 512}) {
    if (th == 256) {
      run<0>("0: 12 MFMA", 256); run<1>("1: 32 exp + 32 mul + 16 fma", 256); run<2>("2: MFMA then VALU", 256);
      run<3>("3: interleaved exp", 256); run<4>("4: MFMA + 84 plain VALU interleaved", 256); run<5>("5: 84 plain VALU", 256);
    } else {
      run<0>("0: 12 MFMA", 512); run<1>("1: 32 exp + 32 mul + 16 fma", 512); run<2>("2: MFMA then VALU", 512);
      run<3>("3: interleaved exp", 512); run<4>("4: MFMA + 84 plain VALU interleaved", 512); run<5>("5: 84 plain VALU", 512);
    }
  }
  return 0;
}
