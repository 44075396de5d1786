// What a FORK costs the stream that records it (round 6): a chain of dependent kernels on one stream, every one followed by "record an event here,
// make the side stream wait for it, launch a kernel there" -- with the event flavours HIP offers and with hipStreamWriteValue32 / WaitValue32.
//   hipcc --offload-arch=gfx950 -O2 -o forkcost forkcost.hip && ./forkcost [iterations] [spin cycles]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void spin(unsigned long long cycles, int* sink) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (sink && threadIdx.x == 1024) *sink = 1;
}
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 200;
  const unsigned long long cyc = argc > 2 ? atoll(argv[2]) : 1000;   // 100 MHz wall clock: 1000 = 10 us
  hipStream_t mainS, side;
  CK(hipStreamCreateWithFlags(&mainS, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  hipEvent_t t0, t1;
  CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  unsigned* flag; CK(hipMalloc(&flag, 4)); CK(hipMemset(flag, 0, 4));
  struct Mode { const char* name; int kind; unsigned flags; };
  const Mode modes[] = {{"no fork (chain only)", 0, 0},
                        {"event, hipEventDisableTiming", 1, hipEventDisableTiming},
                        {"event, DisableTiming | ReleaseToDevice", 1, hipEventDisableTiming | hipEventReleaseToDevice},
                        {"event, DisableTiming | ReleaseToSystem", 1, hipEventDisableTiming | hipEventReleaseToSystem},
                        {"event, default (timing enabled)", 1, hipEventDefault},
                        {"hipStreamWriteValue32 / WaitValue32", 2, 0},
                        {"fork + JOIN per kernel (main also waits for the side kernel's event)", 3, hipEventDisableTiming}};
  for (const Mode& m : modes) {
    std::vector<hipEvent_t> ev(n), ev2(n);
    if (m.kind == 1 || m.kind == 3) for (int i = 0; i < n; ++i) { CK(hipEventCreateWithFlags(&ev[i], m.flags)); CK(hipEventCreateWithFlags(&ev2[i], m.flags)); }
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipMemsetAsync(flag, 0, 4, mainS));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(t0, mainS));
      for (int i = 0; i < n; ++i) {
        hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, mainS, cyc, nullptr);
        if (m.kind == 1 || m.kind == 3) {
          CK(hipEventRecord(ev[i], mainS));
          CK(hipStreamWaitEvent(side, ev[i], 0));
          hipLaunchKernelGGL(spin, dim3(8), dim3(256), 0, side, cyc / 2, nullptr);
          if (m.kind == 3) { CK(hipEventRecord(ev2[i], side)); CK(hipStreamWaitEvent(mainS, ev2[i], 0)); }
        } else if (m.kind == 2) {
          CK(hipStreamWriteValue32(mainS, flag, (unsigned)(i + 1), 0));
          CK(hipStreamWaitValue32(side, flag, (unsigned)(i + 1), hipStreamWaitValueGte, 0xffffffffu));
          hipLaunchKernelGGL(spin, dim3(8), dim3(256), 0, side, cyc / 2, nullptr);
        }
      }
      CK(hipEventRecord(t1, mainS));
      CK(hipDeviceSynchronize());
      float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1));
      if (rep == 1) printf("%-75s %7.2f us per kernel of the chain\n", m.name, ms * 1e3 / n);
    }
    if (m.kind == 1 || m.kind == 3) for (int i = 0; i < n; ++i) { CK(hipEventDestroy(ev[i])); CK(hipEventDestroy(ev2[i])); }
  }
  return 0;
}
