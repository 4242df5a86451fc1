// Diagnostic (not part of the product): what does v_mfma_f32_16x16x4_f32 sustain on this chip
//  (a) from registers only, (b) with the pair-GEMM's LDS fragment reads in the loop?
// hipcc --offload-arch=gfx950 -O3 -Wno-unused-result tools/mfma_peak.hip -o tools/mfma_peak && tools/mfma_peak   (binary is git-ignored)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, unsigned long long *clk) {
  __shared__ __attribute__((aligned(16))) float At[128 * 36], Bt[32 * 132], At2[128 * 36], Bt2[32 * 132];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4, wr = wave >> 1, wc = wave & 1;
  for (int e = tid; e < 128 * 36; e += 256) At[e] = (float)(e % 7) * 0.01f;
  for (int e = tid; e < 32 * 132; e += 256) Bt[e] = (float)(e % 5) * 0.01f;
  __syncthreads();
  f32x4 acc[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
  float a[4][4], b[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { a[i][j] = lane * 0.001f + i + j; b[i][j] = lane * 0.002f - i + j; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    for (int j = 0; j < 32; j += 16) {
      if (MODE == 1 || MODE == 3) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const float4 v = *(const float4 *)&At[((wr * 4 + mi) * 16 + r16) * 36 + j + 4 * g];
          a[mi][0] = v.x; a[mi][1] = v.y; a[mi][2] = v.z; a[mi][3] = v.w;
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const float *bp = &Bt[(j + 4 * g) * 132 + (wc * 4 + ni) * 16 + r16];
          b[ni][0] = bp[0]; b[ni][1] = bp[132]; b[ni][2] = bp[264]; b[ni][3] = bp[396];
        }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi][s], b[ni][s], acc[mi][ni], 0, 0, 0);
    }
    if (MODE == 2) __syncthreads();
    if (MODE == 3) {   // the pair GEMM's full LDS pipeline: 8 x 16-byte stores per thread + one barrier per step
      float *Aw = At2 + (it & 1) * 0, *Bw = Bt2;
#pragma unroll
      for (int q = 0; q < 4; ++q) *(f32x4 *)&Aw[((tid >> 3) + 32 * q) * 36 + ((tid & 7) << 2)] = acc[q][0];
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int e = tid + q * 256; *(f32x4 *)&Bw[(e >> 5) * 132 + ((e & 31) << 2)] = acc[q][1]; }
      __syncthreads();
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
  out[blockIdx.x * 256 + tid] = s;
  if (blockIdx.x == 0 && tid == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE>
void run(const char *name, int wgs, int iters = 4000) {
  float *out; unsigned long long *clk, h[2];
  hipMalloc(&out, wgs * 256 * 4); hipMalloc(&clk, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<wgs, 256>>>(out, 100, clk);
  hipEventRecord(e0);
  k<MODE><<<wgs, 256>>>(out, iters, clk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  double flops = (double)wgs * 4 * iters * 2 * 64 * 2048.0;
  printf("%-28s wgs=%4d iters=%4d  %8.1f us  %.1f TF/s   in-kernel clock %.2f GHz\n", name, wgs, iters, ms * 1e3, flops / ms / 1e9, (double)h[0] / h[1] * 0.1);
  hipFree(out); hipFree(clk);
}

int main() {
  run<0>("registers only, 1 WG/CU", 256);
  run<0>("registers only, 2 WG/CU", 512);
  run<1>("LDS fragments, 1 WG/CU", 256);
  run<1>("LDS fragments, 2 WG/CU", 512);
  run<2>("regs + barrier/step, 2 WG/CU", 512);
  run<3>("LDS frag+store+barrier, 1/CU", 256);
  run<3>("LDS frag+store+barrier, 2/CU", 512);
  run<3>("same, 8 steps", 256, 8);
  run<3>("same, 8 steps", 434, 8);
  run<3>("same, 8 steps", 512, 8);
  run<3>("same, 4 steps", 2246, 4);
  run<1>("LDS frag only, 8 steps", 434, 8);
  run<0>("regs only, 8 steps", 434, 8);
  run<0>("regs only, 8 steps", 256, 8);
  return 0;
}
