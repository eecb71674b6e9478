// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.  wave = 64 lanes throughout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "aldm_hip.h"

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define ALDM_WAVE 64

void aldm_set_error(const char* fmt, ...);

#define ALDM_CHECK_ARG(cond, ...)            \
  do {                                       \
    if (!(cond)) {                           \
      aldm_set_error(__VA_ARGS__);           \
      return ALDM_E_ARG;                     \
    }                                        \
  } while (0)

static inline int aldm_launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    aldm_set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return ALDM_OK;
}

// One-time hipFuncSetAttribute(MaxDynamicSharedMemorySize) per kernel AND per device: the attribute lives with the device's
// copy of the code object, so a process that launches on a second GPU must set it there too.  `done` is the call site's static
// bit mask (bit = device ordinal; ordinals >= 64 set the attribute on every call, which is merely slower).
static inline int aldm_set_max_lds(const void* kern, int bytes, unsigned long long* done, const char* what) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 64;
  if (dev < 64 && ((*done >> dev) & 1ull)) return ALDM_OK;
  hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    aldm_set_error("%s: hipFuncSetAttribute(%d B of LDS): %s", what, bytes, hipGetErrorString(e));
    return (int)e;
  }
  if (dev < 64) *done |= 1ull << dev;
  return ALDM_OK;
}

// n / d for a launch-constant d (round-up magic number, exact for 0 <= n < 2^31): a hardware integer division is ~40 instructions
struct AldmDiv { unsigned mul, shift; };
__device__ __forceinline__ int aldm_div(int n, AldmDiv d) {
  return (int)(((unsigned long long)__umulhi((unsigned)n, d.mul) + (unsigned)n) >> d.shift);
}
static inline AldmDiv aldm_make_div(unsigned d) {
  AldmDiv f;
  unsigned s = 0;
  while ((1ull << s) < d) ++s;
  f.shift = s;
  f.mul = (unsigned)((((1ull << s) - d) << 32) / d + 1);
  return f;
}

// sigmoid / SiLU through v_exp_f32 + v_rcp_f32 (1 ulp): an IEEE fp32 division costs ~10 more VALU instructions per element
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * x)); }
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, i.e. exact for fp32-accumulated bf16 activations): 18 VALU instructions
// against 38 for ocml's erff -- the GEGLU epilogue evaluates it 8.2 M times per feed-forward GEMM at the widest level.
__device__ __forceinline__ float erf_as_f(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  p *= t;
  const float e = __builtin_amdgcn_exp2f(-1.44269504088896340736f * ax * ax);
  return copysignf(fmaf(-p, e, 1.0f), x);
}
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erf_as_f(x * 0.70710678118654752440f)); }

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  switch (act) {
    case ALDM_ACT_SILU: return silu_f(v);
    case ALDM_ACT_LRELU: return v > 0.f ? v : v * slope;
    case ALDM_ACT_TANH: return tanhf(v);
    case ALDM_ACT_GELU: return gelu_erf_f(v);
    default: return v;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline long long cdivll(long long a, long long b) { return (a + b - 1) / b; }
// Touch every 64-byte line of the kernel-argument segment in ONE round trip, as the kernel's first statement.
// The compiler fetches a by-value argument struct lazily, in as many dependent s_load rounds as its SGPR budget suggests (five in
// igemm_pipe_kernel), and in a replayed graph the segment is cold in the scalar cache AND in L2 at every launch: each round that
// meets a new line costs a full memory round trip (~1 us) before the first operand load can even be issued.  After this, all of
// them hit the scalar cache.
template <int BYTES>
__device__ __forceinline__ void aldm_touch_kernargs() {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(BYTES <= 640, "ten 64-byte lines");
  // (one asm block: left to the compiler the loads come out as two or three waited groups)
  constexpr int LAST = (BYTES - 1) / 64 * 64;
#define ALDM_KA_OFF(i) ((i) * 64 < LAST ? (i) * 64 : LAST)
  const void* ka = __builtin_amdgcn_kernarg_segment_ptr();
  unsigned t0, t1, t2, t3, t4, t5, t6, t7, t8, t9;
  asm volatile(
      "s_load_dword %0, %10, %11\n\ts_load_dword %1, %10, %12\n\ts_load_dword %2, %10, %13\n\ts_load_dword %3, %10, %14\n\t"
      "s_load_dword %4, %10, %15\n\ts_load_dword %5, %10, %16\n\ts_load_dword %6, %10, %17\n\ts_load_dword %7, %10, %18\n\t"
      "s_load_dword %8, %10, %19\n\ts_load_dword %9, %10, %20\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "=&s"(t7), "=&s"(t8), "=&s"(t9)
      : "s"(ka), "n"(ALDM_KA_OFF(0)), "n"(ALDM_KA_OFF(1)), "n"(ALDM_KA_OFF(2)), "n"(ALDM_KA_OFF(3)), "n"(ALDM_KA_OFF(4)),
        "n"(ALDM_KA_OFF(5)), "n"(ALDM_KA_OFF(6)), "n"(ALDM_KA_OFF(7)), "n"(ALDM_KA_OFF(8)), "n"(ALDM_KA_OFF(9))
      : "memory");
#undef ALDM_KA_OFF
#endif
}

// bf16 x 8 -> OCP e4m3 x 8 (the fp8 operand form of BASELINE config 5; v_cvt_pk_fp8_f32), and two dwords -> one 64-bit MFMA operand
__device__ __forceinline__ uint2 bf16x8_to_fp8(const bf16x8 v) {
  const uint4 u = __builtin_bit_cast(uint4, v);
  const unsigned w[4] = {u.x, u.y, u.z, u.w};
  int o[2] = {0, 0};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float lo = __uint_as_float(w[i] << 16), hi = __uint_as_float(w[i] & 0xffff0000u);
    if (i & 1) o[i >> 1] = __builtin_amdgcn_cvt_pk_fp8_f32(lo, hi, o[i >> 1], true);
    else o[i >> 1] = __builtin_amdgcn_cvt_pk_fp8_f32(lo, hi, o[i >> 1], false);
  }
  return make_uint2((unsigned)o[0], (unsigned)o[1]);
}
__device__ __forceinline__ long pack2(unsigned lo, unsigned hi) { return (long)(((unsigned long long)hi << 32) | lo); }

// Warm the NEXT launch's kernel-argument segment in this XCD's L2.  In a captured hipGraph (and in the eager kernarg ring) the segments of
// consecutive launches sit back to back in device memory (tools/micro/kernarg_probe.hip: 640-byte stride for a 584-byte struct, fixed
// addresses across replays), and every launch starts by fetching its own, L2-cold, segment (aldm_touch_kernargs: one round trip, but a
// full one).  Called once per workgroup AFTER the main loop, by threads t = 0 .. 23: lane t reads one dword of the t-th 64-byte line
// behind this launch's segment -- 1.5 KB, the next one to three segments -- so that the lines arrive under this launch's epilogue and are
// still resident when the next launch touches them.  Never crosses out of the 4 KB page of this launch's own segment (the pool's end
// is the only place where the bytes behind a segment may be unmapped); the values are discarded.
template <int BYTES>
__device__ __forceinline__ void aldm_prefetch_next_kernargs(int t) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (t < 24) {
    unsigned long long kau = reinterpret_cast<unsigned long long>(__builtin_amdgcn_kernarg_segment_ptr());
    asm volatile("" : "+s"(kau));                            // (keep the segment pointer a scalar of its own: the per-lane address below must
                                                             //  not drag the pointer other code hands to scalar loads into VGPRs)
    const char* ka = reinterpret_cast<const char*>(kau);
    const char* a = ka + ((BYTES + 127) & ~127) + t * 64;
    if (((reinterpret_cast<unsigned long long>(a) ^ reinterpret_cast<unsigned long long>(ka)) >> 12) != 0) a = ka;
    const unsigned v = *reinterpret_cast<const volatile unsigned*>(a);
    (void)v;
  }
#endif
}
