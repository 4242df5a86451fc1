// What does a hand-over between workgroups cost on this part, against the launch boundary it would replace?
//   (1) launch boundary: dependent tiny kernels back to back on one stream, drained time per launch;
//   (2) grid barrier over workgroups on ALL eight XCDs (arrive counter in HBM-coherent memory, device-scope fences);
//   (3) the same barrier over the workgroups of ONE XCD only (workgroup i of a launch goes to XCD i % 8: the launch is 8 x as wide
//       and only the workgroups with blockIdx.x % 8 == 0 take part - they share one L2);
//   (4) (2) and (3) with a 16 KB partial-sum exchange behind the barrier (every workgroup writes 64 floats, reads all of them back):
//       the shape of "BatchNorm statistics inside the producing kernel".
// Diagnostic, not in the library.  build: hipcc --offload-arch=gfx950 -O2 -o /tmp/xcd_handover tools/experiments/xcd_handover.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void tiny(float *p) {
  if (threadIdx.x == 0 && p[0] < 0.f) p[1] = 1.f;
}

// iters barriers among the `n` participating workgroups (all co-resident: the host checks n against the CU count); bounded spin
__global__ __launch_bounds__(256) void barrier_loop(unsigned *counter, int iters, int stride, unsigned n, float *xchg, int exchange,
                                                    unsigned *timeouts, int flagged) {
  if (blockIdx.x % stride != 0) return;
  const unsigned me = blockIdx.x / stride;
  float acc = 0.f;
  __shared__ int gave_up;
  if (threadIdx.x == 0) gave_up = 0;
  for (int it = 0; it < iters; ++it) {
    if (exchange && threadIdx.x < 64) xchg[(size_t)(it & 1) * n * 64 + me * 64 + threadIdx.x] = acc + (float)threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      const unsigned want = (unsigned)(it + 1) * n;
      const unsigned *watch = counter;
      if (flagged) {
        // the last workgroup to arrive publishes the round in a line of its own: the waiters poll a line nobody does atomics on
        watch = counter + 16;
        if (__hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1 == want)
          __hip_atomic_store(counter + 16, want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
      int spins = 0;
      while (__hip_atomic_load(watch, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 22)) {          // never taken on a healthy run: the grid always drains
          atomicAdd(timeouts, 1u);
          gave_up = 1;
          break;
        }
      }
      __threadfence();
    }
    __syncthreads();
    if (gave_up) return;                  // (uniform: the whole workgroup leaves; its peers run into their own limit once and leave too)
    if (exchange) {                       // every workgroup adds all partials, fixed order (what a fused finish would do)
      float s = 0.f;
      const float *src = xchg + (size_t)(it & 1) * n * 64;
      for (unsigned w = threadIdx.x >> 6; w < n; w += 4) s += __builtin_nontemporal_load(src + w * 64 + (threadIdx.x & 63));
      acc += s * 1e-9f;
    }
  }
  if (exchange && threadIdx.x == 0 && acc == 12345.f) xchg[0] = acc;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("%s, %d CUs\n", prop.name, cus);
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  float *buf;
  CK(hipMalloc(&buf, 1 << 22));
  CK(hipMemset(buf, 0, 1 << 22));
  unsigned *counter, *timeouts;
  CK(hipMalloc(&counter, 256));
  timeouts = counter + 32;

  // (1)
  for (int i = 0; i < 2000; ++i) tiny<<<1, 64, 0, s>>>(buf);
  CK(hipStreamSynchronize(s));
  const int N = 20000;
  double t0 = now();
  for (int i = 0; i < N; ++i) tiny<<<1, 64, 0, s>>>(buf);
  CK(hipStreamSynchronize(s));
  printf("(1) launch boundary: %.2f us per dependent tiny launch (drained, one stream)\n", (now() - t0) / N * 1e6);
  t0 = now();
  for (int i = 0; i < N; ++i) tiny<<<256, 256, 0, s>>>(buf);
  CK(hipStreamSynchronize(s));
  printf("(1) launch boundary: %.2f us per dependent 256-workgroup launch\n", (now() - t0) / N * 1e6);

  struct Case { const char *name; int grid, stride; unsigned n; int exchange; };
  const std::vector<Case> cases = {
      {"(2) barrier, 256 workgroups on all XCDs", 256, 1, 256, 0},
      {"(2) barrier, 64 workgroups on all XCDs", 64, 1, 64, 0},
      {"(3) barrier, 32 workgroups of ONE XCD", 256, 8, 32, 0},
      {"(3) barrier, 64 workgroups of ONE XCD (2 per CU)", 512, 8, 64, 0},
      {"(4) barrier + 16-64 KB exchange, 256 workgroups on all XCDs", 256, 1, 256, 1},
      {"(4) barrier + exchange, 64 workgroups on all XCDs", 64, 1, 64, 1},
      {"(4) barrier + exchange, 32 workgroups of ONE XCD", 256, 8, 32, 1},
      {"(4) barrier + exchange, 64 workgroups of ONE XCD", 512, 8, 64, 1},
  };
  const int iters = 2000;
  for (int flagged = 0; flagged < 2; ++flagged)
  for (const Case &c : cases) {
    if ((int)c.n > cus) {
      printf("%s: skipped (%u workgroups > %d CUs)\n", c.name, c.n, cus);
      continue;
    }
    double best = 1e30;
    unsigned to = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemsetAsync(counter, 0, 256, s));
      CK(hipStreamSynchronize(s));
      t0 = now();
      barrier_loop<<<c.grid, 256, 0, s>>>(counter, iters, c.stride, c.n, buf, c.exchange, timeouts, flagged);
      CK(hipStreamSynchronize(s));
      best = std::min(best, now() - t0);
      CK(hipMemcpy(&to, timeouts, 4, hipMemcpyDeviceToHost));
    }
    printf("%s%s: %.2f us per round%s\n", flagged ? "[flag line] " : "[counter polled] ", c.name, best / iters * 1e6, to ? "  (SPIN LIMIT HIT: figure invalid)" : "");
  }
  return 0;
}
