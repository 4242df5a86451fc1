// Host cost of one kernel launch on this box: <<<>>> against hipModuleLaunchKernel on a cached hipFunction_t
// (hipGetFuncBySymbol), with the arguments of a typical library kernel (10 pointers / sizes).  Diagnostic, not in the library.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/launch_cost tools/experiments/launch_cost.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>

__global__ void k10(const float *a, const float *b, float *c, const int *d, int64_t n, int e, int f, float g, float *h, int *i) {
  if (n < 0) c[0] = a[0] + b[0] + d[0] + e + f + g + h[0] + i[0];
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  float *buf;
  hipMalloc(&buf, 1 << 20);
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const int N = 20000;
  const float *a = buf, *b = buf;
  float *c = buf, *h = buf;
  const int *d = (const int *)buf;
  int *ip = (int *)buf;
  int64_t n = 0;
  int e = 1, f = 2;
  float g = 3.f;
  for (int rep = 0; rep < 2; ++rep) {
    for (int i = 0; i < 1000; ++i) k10<<<1, 64, 0, s>>>(a, b, c, d, n, e, f, g, h, ip);
    hipStreamSynchronize(s);
    double t0 = now();
    for (int i = 0; i < N; ++i) k10<<<1, 64, 0, s>>>(a, b, c, d, n, e, f, g, h, ip);
    double t1 = now();
    hipStreamSynchronize(s);
    double t2 = now();
    printf("<<<>>>                  issue %.2f us / launch, drained %.2f us / launch\n", (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6);

    hipFunction_t fn;
    hipGetFuncBySymbol(&fn, (const void *)k10);
    void *args[] = {&a, &b, &c, &d, &n, &e, &f, &g, &h, &ip};
    for (int i = 0; i < 1000; ++i) hipModuleLaunchKernel(fn, 1, 1, 1, 64, 1, 1, 0, s, args, nullptr);
    hipStreamSynchronize(s);
    t0 = now();
    for (int i = 0; i < N; ++i) hipModuleLaunchKernel(fn, 1, 1, 1, 64, 1, 1, 0, s, args, nullptr);
    t1 = now();
    hipStreamSynchronize(s);
    t2 = now();
    printf("hipModuleLaunchKernel   issue %.2f us / launch, drained %.2f us / launch\n", (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6);

    // a short burst (the queue never fills): what a host-bound step sees
    hipStreamSynchronize(s);
    double acc = 0;
    for (int r = 0; r < 200; ++r) {
      t0 = now();
      for (int i = 0; i < 20; ++i) k10<<<1, 64, 0, s>>>(a, b, c, d, n, e, f, g, h, ip);
      t1 = now();
      acc += t1 - t0;
      hipStreamSynchronize(s);
    }
    printf("<<<>>> bursts of 20     issue %.2f us / launch\n", acc / (200 * 20) * 1e6);
    // launches that alternate with event records (what a profiled step does)
    hipEvent_t ev;
    hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    t0 = now();
    for (int i = 0; i < N; ++i) {
      k10<<<1, 64, 0, s>>>(a, b, c, d, n, e, f, g, h, ip);
      if ((i & 7) == 0) hipEventRecord(ev, s);
    }
    t1 = now();
    hipStreamSynchronize(s);
    printf("<<<>>> + event / 8      issue %.2f us / launch\n", (t1 - t0) / N * 1e6);
  }
  return 0;
}
