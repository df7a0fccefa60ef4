// hbm_pattern_probe.hip - round 6: what does the ACCESS PATTERN of the fused norm -> Linear prologue cost in HBM bandwidth?
//
// The prologue of dss_lnlinear_k384 reads (and writes back) the fp32 residual stream x [M, 384] in UNITS of 32 rows x 32 columns:
// one wave instruction moves 8 rows x 128 bytes - eight 128-byte lines that are 1 536 bytes apart -, and a row's twelve lines are
// requested over twelve consecutive units.  A streaming copy touches a DRAM page once; this pattern touches it once per line.
// The probe moves the same bytes three ways and prints TB/s:
//   rows      x row-major [M][384] fp32, walked in 32-row x 32-column units as the kernel does
//   strips    x strip-major [12][M][32] fp32 (every 32-column strip of all rows contiguous: a unit is 4 KB contiguous)
//   stream    a plain contiguous sweep (what the chip can do)
// each as read-only and as read + write-back (the kernel's x += r form), in 256-thread workgroups of 4 waves x 64 rows like the
// kernel's, two per CU by the LDS they declare.
//
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/probes/hbm_pattern_probe.hip -o scripts/probes/hbm_pattern_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
static constexpr int D = 384, NCB = D / 32;

// MODE 0: rows (row-major, strided units)   1: strips (strip-major, contiguous units)   2: stream (contiguous sweep of the block)
template <int MODE, int WRITE>     // WRITE 0: read only   1: read + write-back   2: write only
__global__ __launch_bounds__(256, 2) void walk_kernel(float* __restrict__ x, float* __restrict__ sink, long M) {
  __shared__ unsigned char pad[80 * 1024];            // two workgroups per CU, like the Linear kernel's
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0 && M < 0) pad[0] = 1;                  // (keeps the allocation)
  const long row0 = (long)blockIdx.x * 256 + wave * 64;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int prow = lane >> 3, pch = lane & 7;         // a wave instruction: 8 rows x 128 B, lane -> (row, 16-byte chunk)
#pragma unroll 1
  for (int t = 0; t < 2; ++t) {
#pragma unroll 1
    for (int cb = 0; cb < NCB; ++cb) {
      f32x4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        long r = row0 + 32 * t + 8 * q + prow;
        if (r >= M) r = M - 1;
        float* p;
        if (MODE == 0) p = x + r * D + 32 * cb + 4 * pch;
        else if (MODE == 1) p = x + (long)cb * M * 32 + r * 32 + 4 * pch;
        else p = x + ((long)blockIdx.x * 256 + wave * 64) * D + ((t * NCB + cb) * 4 + q) * 256 + 4 * lane;   // 1 KB contiguous per instruction
        if (WRITE != 2) v[q] = *reinterpret_cast<const f32x4*>(p);
        else v[q] = f32x4{(float)q, 1.f, 2.f, (float)cb};
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc += v[q];
        if (WRITE) {
          long r = row0 + 32 * t + 8 * q + prow;
          if (r >= M) r = M - 1;
          float* p;
          if (MODE == 0) p = x + r * D + 32 * cb + 4 * pch;
          else if (MODE == 1) p = x + (long)cb * M * 32 + r * 32 + 4 * pch;
          else p = x + ((long)blockIdx.x * 256 + wave * 64) * D + ((t * NCB + cb) * 4 + q) * 256 + 4 * lane;
          *reinterpret_cast<f32x4*>(p) = v[q] + 1.0f;
        }
      }
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

// The fused norm -> Linear kernel's traffic replayed by a kernel that does nothing else: per workgroup of 4 waves x 64 rows the
// PROLOGUE (per 32 x 32 unit: x f32 in, the pending branch output r f16 in - one 32 x 64 tile per two units -, x f32 back) and then
// the OUTPUT (N f16 columns per row, 64-column groups of full 128-byte lines, 8 rows per instruction), in the kernel's order, with the
// kernel's residency (two workgroups per CU).  NOUT = 1152 (qkv) / 1536 (fc1).
template <int NOUT>
__global__ __launch_bounds__(256, 2) void replay_kernel(float* __restrict__ x, const unsigned short* __restrict__ r, unsigned short* __restrict__ out,
                                                        float* __restrict__ sink, long M) {
  __shared__ unsigned char pad[80 * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0 && M < 0) pad[0] = 1;
  const long row0 = (long)blockIdx.x * 256 + wave * 64;
  const int prow = lane >> 3, pch = lane & 7;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int t = 0; t < 2; ++t) {
#pragma unroll 1
    for (int cb = 0; cb < NCB; ++cb) {
      f32x4 v[4], rv[2];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        long rr = row0 + 32 * t + 8 * q + prow; if (rr >= M) rr = M - 1;
        v[q] = *reinterpret_cast<const f32x4*>(x + rr * D + 32 * cb + 4 * pch);
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {            // half of the 32 x 64 f16 tile per unit: 16 rows x 64 B per instruction
        long rr = row0 + 32 * t + 16 * q + (lane >> 2); if (rr >= M) rr = M - 1;
        rv[q] = *reinterpret_cast<const f32x4*>(r + rr * D + 32 * cb + 8 * (lane & 3));
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        long rr = row0 + 32 * t + 8 * q + prow; if (rr >= M) rr = M - 1;
        acc += rv[q & 1];
        *reinterpret_cast<f32x4*>(x + rr * D + 32 * cb + 4 * pch) = v[q] + rv[q & 1];
      }
    }
  }
#pragma unroll 1
  for (int g = 0; g < NOUT / 64; ++g) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      long rr = row0 + 8 * q + prow; if (rr >= M) rr = M - 1;
      *reinterpret_cast<f32x4*>(out + rr * NOUT + 64 * g + 8 * pch) = acc + (float)g;
    }
  }
  if (acc[0] == 12345.678f) sink[0] = acc[1];
}

template <int NOUT> static double run_replay(float* x, unsigned short* r, unsigned short* out, float* sink, long M, int reps, double* ms_out) {
  const int grid = (int)((M + 255) / 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((replay_kernel<NOUT>), dim3(grid), dim3(256), 0, 0, x, r, out, sink, M);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int i = 0; i < reps; ++i) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((replay_kernel<NOUT>), dim3(grid), dim3(256), 0, 0, x, r, out, sink, M);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  *ms_out = best;
  return (double)M * (D * 10.0 + NOUT * 2.0) / best / 1e9;
}

template <int MODE, int WRITE> static double run(float* x, float* sink, long M, int reps) {
  const int grid = (int)((M + 255) / 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((walk_kernel<MODE, WRITE>), dim3(grid), dim3(256), 0, 0, x, sink, M);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int i = 0; i < reps; ++i) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((walk_kernel<MODE, WRITE>), dim3(grid), dim3(256), 0, 0, x, sink, M);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double bytes = (double)M * D * 4 * (WRITE == 1 ? 2 : 1);
  return bytes / best / 1e9;   // TB/s
}

int main(int argc, char** argv) {
  const long M = argc > 1 ? atol(argv[1]) : 2228173;
  float *x, *sink;
  if (hipMalloc(&x, (size_t)M * D * 4 + (1 << 20)) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(x, 0, (size_t)M * D * 4);
  printf("x [%ld, %d] fp32 = %.2f GB; 256-thread workgroups of 4 waves x 64 rows, two per CU; TB/s, best of 8\n", M, D, M * D * 4.0 / 1e9);
  printf("                       read only    read + write-back    write only\n");
  printf("rows   (as the kernel)   %6.2f        %6.2f           %6.2f\n", run<0, 0>(x, sink, M, 8), run<0, 1>(x, sink, M, 8), run<0, 2>(x, sink, M, 8));
  printf("strips (strip-major x)   %6.2f        %6.2f           %6.2f\n", run<1, 0>(x, sink, M, 8), run<1, 1>(x, sink, M, 8), run<1, 2>(x, sink, M, 8));
  printf("stream (contiguous)      %6.2f        %6.2f           %6.2f\n", run<2, 0>(x, sink, M, 8), run<2, 1>(x, sink, M, 8), run<2, 2>(x, sink, M, 8));
  unsigned short *r, *out;
  if (hipMalloc(&r, (size_t)M * D * 2 + (1 << 20)) != hipSuccess || hipMalloc(&out, (size_t)M * 1536 * 2 + (1 << 20)) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(r, 0, (size_t)M * D * 2);
  double ms_q, ms_f;
  const double tq = run_replay<1152>(x, r, out, sink, M, 8, &ms_q), tf = run_replay<1536>(x, r, out, sink, M, 8, &ms_f);
  printf("the fused kernel's own traffic, replayed (x f32 in + r f16 in + x f32 back + N f16 columns out, the kernel's order and residency):\n");
  printf("  norm1 -> qkv  (N = 1152): %6.2f TB/s  %.3f ms per launch of %ld rows\n", tq, ms_q, M);
  printf("  norm2 -> fc1  (N = 1536): %6.2f TB/s  %.3f ms\n", tf, ms_f);
  return 0;
}
