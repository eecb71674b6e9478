// How much vector work hides under MFMAs on gfx950, with the instruction stream pinned by inline asm (the compiler cannot re-cluster it)?
//   A: one wave per SIMD, blocks of [1 x v_mfma_f32_32x32x16_bf16 + n x v_fma_f32]  (n = 0..12): cycles per block
//   B: the same with [1 MFMA + n x v_exp_f32]
//   C: two waves per SIMD, the SAME stream in both (in phase)
//   D: two waves per SIMD in OPPOSITE phases: waves 0-3 run [16 MFMA | 96 fma], waves 4-7 run [96 fma | 16 MFMA], no barrier between the
//      phases (free-running), against E: both halves run [16 MFMA | 96 fma] (in phase) and F: the fine interleave [1 MFMA + 6 fma] x 16.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_gap.hip -o /tmp/mfma_valu_gap ; prints shader cycles (s_memtime) per block.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define MFMA(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(k1), "v"(k2))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))

template <int N, bool USE_EXP>
__device__ __forceinline__ void block(f32x16& acc, const bf16x8& a, const bf16x8& b, float (&x)[12], float k1, float k2) {
  MFMA(acc);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (USE_EXP) EXP(x[i % 12]); else FMA(x[i % 12]);
  }
}

// MODE 0: [1 MFMA + N fillers] x 16 per iteration.  MODE 1: [16 MFMA | 16 N fillers] (phases).  MODE 2: as 1, but waves 4.. run the phases
// in the opposite order.
template <int N, bool USE_EXP, int MODE>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(0.5f + i); }
  float x[12];
  for (int i = 0; i < 12; ++i) x[i] = threadIdx.x * 1e-3f + i * 1e-2f;
  const float k1 = 0.9999f, k2 = 0.0001f;
  const bool flip = MODE == 2 && (threadIdx.x >> 8);      // waves 4-7 (the second wave of each SIMD)
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int m = 0; m < 16; ++m) block<N, USE_EXP>(acc[m & 3], a, b, x, k1, k2);
    } else {
      if (!flip) {
#pragma unroll
        for (int m = 0; m < 16; ++m) MFMA(acc[m & 3]);
#pragma unroll
        for (int i = 0; i < 16 * N; ++i) { if (USE_EXP) EXP(x[i % 12]); else FMA(x[i % 12]); }
      } else {
#pragma unroll
        for (int i = 0; i < 16 * N; ++i) { if (USE_EXP) EXP(x[i % 12]); else FMA(x[i % 12]); }
#pragma unroll
        for (int m = 0; m < 16; ++m) MFMA(acc[m & 3]);
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i];
  for (int i = 0; i < 12; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  if (threadIdx.x == 256 && blockIdx.x == 0) cyc[1] = t1 - t0;
}

template <int N, bool USE_EXP, int MODE>
void run(const char* what, int threads) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 16);
  hipMemset(cyc, 0, 16);
  const int iters = 1000;
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<N, USE_EXP, MODE>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long c[2]; hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
  printf("%-26s n=%2d %s waves/SIMD %d: %7.1f cycles per [MFMA + n fillers] (wave 0)", what, N, USE_EXP ? "exp" : "fma", threads / 256,
         (double)c[0] / iters / 16);
  if (threads == 512) printf(", %7.1f (wave 4)", (double)c[1] / iters / 16);
  printf("\n");
  hipFree(out); hipFree(cyc);
}

int main() {
  printf("A: one wave per SIMD, fine interleave, v_fma fillers\n");
  run<0, false, 0>("interleaved", 256); run<2, false, 0>("interleaved", 256); run<4, false, 0>("interleaved", 256);
  run<5, false, 0>("interleaved", 256); run<6, false, 0>("interleaved", 256); run<7, false, 0>("interleaved", 256);
  run<8, false, 0>("interleaved", 256); run<12, false, 0>("interleaved", 256);
  printf("B: one wave per SIMD, fine interleave, v_exp fillers\n");
  run<1, true, 0>("interleaved", 256); run<2, true, 0>("interleaved", 256); run<3, true, 0>("interleaved", 256); run<4, true, 0>("interleaved", 256);
  printf("   phases [16 MFMA | 16 n fillers], one wave per SIMD (no overlap possible: the sum)\n");
  run<6, false, 1>("phases", 256); run<12, false, 1>("phases", 256); run<3, true, 1>("phases", 256);
  printf("C: two waves per SIMD, same stream in both, fine interleave (per wave; two blocks retire per printed cycles)\n");
  run<0, false, 0>("interleaved", 512); run<3, false, 0>("interleaved", 512); run<6, false, 0>("interleaved", 512); run<12, false, 0>("interleaved", 512);
  run<3, true, 0>("interleaved", 512);
  printf("E: two waves per SIMD, phases IN phase\n");
  run<6, false, 1>("phases in phase", 512); run<12, false, 1>("phases in phase", 512); run<24, false, 1>("phases in phase", 512); run<3, true, 1>("phases in phase", 512);
  printf("D: two waves per SIMD, phases in OPPOSITE phase (waves 4-7 start with the fillers)\n");
  run<6, false, 2>("phases opposite", 512); run<12, false, 2>("phases opposite", 512); run<24, false, 2>("phases opposite", 512); run<3, true, 2>("phases opposite", 512);
  return 0;
}
