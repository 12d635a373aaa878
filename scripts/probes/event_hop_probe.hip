// What does a cross-stream hop cost the MAIN stream on this runtime?  Main stream: A -> B -> D; a small piece of work C
// is forked behind A onto a side stream (record e0 on main, side waits e0, C, record e1 on side) and joined in front of
// D (main waits e1).  C = a four-workgroup kernel, or a 1-MB device-to-device hipMemcpyAsync (what RCCL issues for a
// collective at world size 1).  Reported: wall per iteration of A B D alone, with C in order on the main stream, and
// with C forked -- B is sized like a fused per-sample launch (one workgroup per CU, ~35 us).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/event_hop_probe scripts/probes/event_hop_probe.hip && /tmp/event_hop_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin(float* p, long long cycles) {
  const long long t0 = wall_clock64();
  float v = p[threadIdx.x];
  while (wall_clock64() - t0 < cycles) v = v * 1.0001f + 1.f;
  p[blockIdx.x * blockDim.x + threadIdx.x] = v;
}

int main() {
  float *buf, *src, *dst;
  CK(hipMalloc(&buf, 512 * 256 * 4)); CK(hipMalloc(&src, 1 << 20)); CK(hipMalloc(&dst, 1 << 20));
  CK(hipMemset(buf, 0, 512 * 256 * 4));
  hipStream_t mainst, side;
  CK(hipStreamCreateWithFlags(&mainst, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreateWithFlags(&e0, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
  const long long us = 100;                 // wall_clock64 ticks at 100 MHz
  auto A = [&]() { hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, mainst, buf, 8 * us); };
  auto B = [&]() { hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, mainst, buf, 35 * us); };
  auto D = [&]() { hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, mainst, buf, 8 * us); };
  for (int mode = 0; mode < 5; ++mode) {
    // 0: A B D   1: A Ck B D in order   2: A [Ck forked] B join D   3: A Cm B D in order   4: A [Cm forked] B join D
    auto C = [&](hipStream_t s) {
      if (mode <= 2) hipLaunchKernelGGL(spin, dim3(4), dim3(256), 0, s, buf + 256 * 256, 4 * us);
      else (void)hipMemcpyAsync(dst, src, 1 << 20, hipMemcpyDeviceToDevice, s);
    };
    double best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
      const int K = 200;
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < K; ++i) {
        A();
        if (mode == 1 || mode == 3) C(mainst);
        if (mode == 2 || mode == 4) { (void)hipEventRecord(e0, mainst); (void)hipStreamWaitEvent(side, e0, 0); C(side); (void)hipEventRecord(e1, side); }
        B();
        if (mode == 2 || mode == 4) (void)hipStreamWaitEvent(mainst, e1, 0);
        D();
      }
      CK(hipDeviceSynchronize());
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6 / K;
      if (rep > 0 && dt < best) best = dt;
    }
    const char* names[5] = {"A B D", "A C(kernel) B D, in order", "A [C(kernel) forked] B join D", "A C(memcpy 1 MB) B D, in order",
                            "A [C(memcpy 1 MB) forked] B join D"};
    printf("%-40s %7.1f us per iteration\n", names[mode], best);
  }
  return 0;
}
