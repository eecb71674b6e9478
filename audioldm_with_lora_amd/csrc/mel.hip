// Log-mel front end for gfx950: waveform -> log(clamp(mel_basis |STFT|, 1e-5)) in one kernel.
//
// Replaces, on the GPU, the reference dataloader's `mel_spectrogram_train` + `pad_spec`
// [REF script/data/datasets.py:301-354, 385-398]: reflect-pad (n_fft - hop)/2, torch.stft(n_fft 1024, hop 160,
// periodic Hann 1024, center=False, onesided) -> magnitude -> 64 Slaney mel filters -> natural log of the value
// clamped at 1e-5 -> [frames, mel], zero-padded / cropped to target_length frames.
//
// One 256-thread workgroup per frame.  The 1024-point transform is a radix-4 Stockham autosort FFT held entirely in LDS
// (5 passes, one butterfly per thread per pass, ping-pong float2 images, twiddles exp(-2 pi i k / 1024) tabulated once
// per workgroup with sincospif); fp32 throughout, as torch.stft is.  The mel projection uses each filter's non-zero
// bin range (triangles are <= 46 bins wide), four lanes per filter.  HBM traffic is the 4 KB of samples per frame
// (overlapping hops hit L2) and 256 B of output: the kernel is latency / LDS bound, not an MFMA candidate.
#include "common.h"

namespace {

constexpr int NFFT = 1024;
constexpr int NBINS = NFFT / 2 + 1;

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

__global__ __launch_bounds__(256) void log_mel_kernel(const float* __restrict__ wav, int T, int hop, int n_frames,
                                                      int target_frames, const float* __restrict__ window,
                                                      const float* __restrict__ mel_basis, const int* __restrict__ mel_range,
                                                      int n_mels, float clamp_min, float* __restrict__ out) {
  __shared__ float2 bufA[NFFT], bufB[NFFT], tw[768];
  __shared__ float mag[NBINS + 3];
  const int tid = threadIdx.x;
  const int frame = blockIdx.x, b = blockIdx.y;
  float* orow = out + ((long long)b * target_frames + frame) * n_mels;
  if (frame >= n_frames) {                       // pad_spec: frames past the clip are zeros (not log(clamp))
    for (int m = tid; m < n_mels; m += 256) orow[m] = 0.f;
    return;
  }
  for (int k = tid; k < 768; k += 256) {
    float s, c;
    sincospif(-(float)k * (2.0f / NFFT), &s, &c);
    tw[k] = make_float2(c, s);
  }
  // windowed frame with the reflect padding folded into the index
  const float* w = wav + (long long)b * T;
  const int pad = (NFFT - hop) / 2;
  for (int i = tid; i < NFFT; i += 256) {
    int src = frame * hop + i - pad;
    src = src < 0 ? -src : src;
    src = src >= T ? 2 * (T - 1) - src : src;
    bufA[i] = make_float2(w[src] * window[i], 0.f);
  }
  __syncthreads();
  float2* x = bufA;
  float2* y = bufB;
#pragma unroll
  for (int pass = 0; pass < 5; ++pass) {
    const int s = 1 << (2 * pass), n = NFFT >> (2 * pass), n1 = n >> 2;
    const int p = tid >> (2 * pass), q = tid & (s - 1);
    const float2 a = x[q + s * p], bb = x[q + s * (p + n1)], c = x[q + s * (p + 2 * n1)], d = x[q + s * (p + 3 * n1)];
    const float2 apc = make_float2(a.x + c.x, a.y + c.y), amc = make_float2(a.x - c.x, a.y - c.y);
    const float2 bpd = make_float2(bb.x + d.x, bb.y + d.y);
    const float2 jbmd = make_float2(-(bb.y - d.y), bb.x - d.x);                    // i * (b - d)
    const int k1 = p * s;                                                          // twiddle exp(-2 pi i p / n)
    y[q + s * (4 * p + 0)] = make_float2(apc.x + bpd.x, apc.y + bpd.y);
    y[q + s * (4 * p + 1)] = cmul(tw[k1], make_float2(amc.x - jbmd.x, amc.y - jbmd.y));
    y[q + s * (4 * p + 2)] = cmul(tw[2 * k1], make_float2(apc.x - bpd.x, apc.y - bpd.y));
    y[q + s * (4 * p + 3)] = cmul(tw[3 * k1], make_float2(amc.x + jbmd.x, amc.y + jbmd.y));
    __syncthreads();
    float2* t = x; x = y; y = t;
  }
  for (int k = tid; k < NBINS; k += 256) mag[k] = sqrtf(x[k].x * x[k].x + x[k].y * x[k].y);
  __syncthreads();
  // four lanes per mel filter over its non-zero bins
  for (int m0 = 0; m0 < n_mels; m0 += 64) {
    const int m = m0 + (tid >> 2), part = tid & 3;
    float acc = 0.f;
    if (m < n_mels) {
      const int lo = mel_range[2 * m], hi = mel_range[2 * m + 1];
      const float* fr = mel_basis + (long long)m * NBINS;
      for (int k = lo + part; k < hi; k += 4) acc += fr[k] * mag[k];
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (m < n_mels && part == 0) orow[m] = logf(fmaxf(acc, clamp_min));
  }
}

}  // namespace

extern "C" int aldm_log_mel(const float* wav, int B, int T, int n_fft, int hop, const float* window, const float* mel_basis,
                            const int* mel_range, int n_mels, int target_frames, float clamp_min, float* out,
                            void* stream) {
  ALDM_CHECK_ARG(wav && window && mel_basis && mel_range && out, "log_mel: null pointer");
  ALDM_CHECK_ARG(n_fft == NFFT, "log_mel: n_fft %d unsupported (the LDS FFT is built for 1024)", n_fft);
  ALDM_CHECK_ARG(B > 0 && hop > 0 && hop <= NFFT && (NFFT - hop) % 2 == 0 && n_mels > 0 && target_frames > 0, "log_mel: bad dims");
  ALDM_CHECK_ARG(T > (NFFT - hop) / 2, "log_mel: %d samples are too few for reflect padding of %d", T, (NFFT - hop) / 2);
  const int padded = T + (NFFT - hop);
  int n_frames = padded >= NFFT ? 1 + (padded - NFFT) / hop : 0;
  if (n_frames > target_frames) n_frames = target_frames;           // pad_spec crops to target_length
  hipLaunchKernelGGL(log_mel_kernel, dim3(target_frames, B), dim3(256), 0, (hipStream_t)stream, wav, T, hop, n_frames,
                     target_frames, window, mel_basis, mel_range, n_mels, clamp_min, out);
  return aldm_launch_status("log_mel");
}
