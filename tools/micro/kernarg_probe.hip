// Where do the kernel-argument segments of a replayed hipGraph's nodes live?  Every launch stores its own segment pointer; the
// program prints them for an eager run and for two replays of a captured graph (round 4: is the NEXT node's segment on the same page?)
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/kernarg_probe.hip -o /tmp/kernarg_probe
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { unsigned long long* out; int idx; int pad[140]; };   // 584 bytes like IgemmDev
__global__ void probe(const Big b) {
  if (threadIdx.x == 0 && blockIdx.x == 0) b.out[b.idx] = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
}
__global__ void small(unsigned long long* out, int idx) {
  if (threadIdx.x == 0 && blockIdx.x == 0) out[idx] = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
}
int main() {
  unsigned long long* d; hipMalloc(&d, 64 * 8); hipMemset(d, 0, 64 * 8);
  hipStream_t st; hipStreamCreate(&st);
  Big b{}; b.out = d;
  for (int i = 0; i < 6; ++i) { b.idx = i; if (i % 3 == 2) small<<<1, 64, 0, st>>>(d, i); else probe<<<256, 64, 0, st>>>(b); }
  hipStreamSynchronize(st);
  unsigned long long h[64]; hipMemcpy(h, d, 64 * 8, hipMemcpyDeviceToHost);
  printf("eager:  "); for (int i = 0; i < 6; ++i) printf("%llx ", h[i]); printf("\n");
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < 12; ++i) { b.idx = 8 + i; if (i % 3 == 2) small<<<1, 64, 0, st>>>(d, 8 + i); else probe<<<256, 64, 0, st>>>(b); }
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int r = 0; r < 2; ++r) {
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipMemcpy(h, d, 64 * 8, hipMemcpyDeviceToHost);
    printf("replay%d: ", r); for (int i = 8; i < 20; ++i) printf("%llx ", h[i]); printf("\n");
    printf("  deltas: "); for (int i = 9; i < 20; ++i) printf("%lld ", (long long)(h[i] - h[i - 1])); printf("\n");
  }
  hipPointerAttribute_t at; if (hipPointerGetAttributes(&at, (void*)h[8]) == hipSuccess) printf("segment memory type %d device %d managed %d\n", (int)at.type, at.device, at.isManaged); else printf("pointer attributes: not a hip allocation\n");
  return 0;
}
