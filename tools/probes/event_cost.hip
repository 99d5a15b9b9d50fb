// What a cross-stream hand-over costs the stream it sits in (round 6, data parallelism: ~45 per step).
// A chain of N kernels (~20 us each) on stream A, and between consecutive kernels one of:
//   none       nothing
//   record     hipEventRecord(e, A); hipStreamWaitEvent(B, e)            (what _on_side_stream does per bucket)
//   wait       hipStreamWaitEvent(A, e_done) on an event of stream B that has long happened   (params_ready per layer)
// each with events created with flags: default|disableTiming (torch's), + hipEventDisableSystemFence, + hipEventReleaseToDevice.
// The host is kept AHEAD of the GPU (a long kernel first), as in the training step.
//   hipcc --offload-arch=gfx950 -O2 -o event_cost event_cost.hip && ./event_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void spin(float* p, int iters) {
  float x = p[threadIdx.x];
  for (int i = 0; i < iters; ++i) x = x * 1.0001f + 0.5f;
  p[threadIdx.x + blockIdx.x * blockDim.x] = x;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  const int N = 200;
  float* buf; CK(hipMalloc(&buf, 256 * 1024 * 4));
  hipStream_t A, B; CK(hipStreamCreate(&A)); CK(hipStreamCreate(&B));
  hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  struct Flag { const char* name; unsigned f; } flags[] = {
    {"disableTiming (torch)", hipEventDisableTiming},
    {"+DisableSystemFence", hipEventDisableTiming | hipEventDisableSystemFence},
    {"+ReleaseToDevice", hipEventDisableTiming | hipEventReleaseToDevice},
  };
  const char* modes[] = {"none", "record", "wait"};
  for (int rep = 0; rep < 2; ++rep)
  for (auto& fl : flags) {
    std::vector<hipEvent_t> ev(N);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, fl.f));
    hipEvent_t done; CK(hipEventCreateWithFlags(&done, fl.f));
    for (int m = 0; m < 3; ++m) {
      CK(hipDeviceSynchronize());
      hipLaunchKernelGGL(spin, dim3(1), dim3(256), 0, B, buf, 10);
      CK(hipEventRecord(done, B));
      CK(hipDeviceSynchronize());
      hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, A, buf, 2000000);      // ~ms: lets the host run ahead
      if (m == 2) {      // `done` is PENDING when the waits are enqueued (it follows the long kernel through stream B), done when they execute
        CK(hipEventRecord(ev[0], A)); CK(hipStreamWaitEvent(B, ev[0], 0));
        hipLaunchKernelGGL(spin, dim3(1), dim3(256), 0, B, buf, 10);
        CK(hipEventRecord(done, B));
        hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, A, buf, 200000);
      }
      CK(hipEventRecord(t0, A));
      for (int i = 0; i < N; ++i) {
        hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, A, buf, 1200);
        if (m == 1) { CK(hipEventRecord(ev[i], A)); CK(hipStreamWaitEvent(B, ev[i], 0)); }
        if (m == 2) { CK(hipStreamWaitEvent(A, done, 0)); }
      }
      CK(hipEventRecord(t1, A));
      CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, t0, t1));
      printf("%-24s %-7s %8.2f us per kernel\n", fl.name, modes[m], ms * 1e3f / N);
    }
    for (auto& e : ev) CK(hipEventDestroy(e));
    CK(hipEventDestroy(done));
  }
  return 0;
}
