// Probe: empirical lane semantics of ds_read_b64_tr_b16 on gfx950 (run once on the GPU box).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void probe(short* out, int mode) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x;
  int idx;                       // half index this lane points at (8-byte aligned)
  if (mode == 0) idx = 4 * l;    // lane l -> halves [4l, 4l+4)
  else {                         // "row-major 4x16 block per 16 lanes, row stride 72 halves": lane i -> row i>>2, col 4*(i&3)
    const int g = l >> 4, i = l & 15;
    idx = g * 1024 + (i >> 2) * 72 + 4 * (i & 3);
  }
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(lds + idx));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  short h[256];
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
  }
  return 0;
}
