// graphgap.hip -- 30-line reproducer for "nodes of a captured HIP graph dispatch ~3x further apart than the same kernels launched eagerly"
// (the training step replayed from one graph has a median inter-kernel gap of 20.8 us against
// 6.6 us eager).  N dependent kernels of ~T us on one stream, (a) launched one by one, (b) captured once and replayed with hipGraphLaunch,
// (c) captured with a second stream forked / joined every F kernels (the shape of the training step: weight-gradient and geometry
// streams).  Prints the wall time per kernel of each mode = kernel time + dispatch gap.
//   hipcc --offload-arch=gfx950 -O2 graphgap.hip -o graphgap && ./graphgap [N=150] [spin_cycles=20000] [F=8] [MB written per kernel=0]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void spin(long long cycles, float* buf, long long floats) {
  // optional memory traffic: every launch rewrites `floats` floats (dirty lines in the XCDs' L2s at the kernel boundary, as the training
  // step's kernels leave them -- a release at a node boundary then has something to write back)
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < floats; i += (long long)gridDim.x * blockDim.x) buf[i] = buf[i] * 1.0001f + 1.f;
  const long long t0 = clock64();
  while (clock64() - t0 < cycles) {}
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 150, F = argc > 3 ? atoi(argv[3]) : 8, reps = 50;
  const long long cyc = argc > 2 ? atoll(argv[2]) : 20000;
  const long long floats = (argc > 4 ? atoll(argv[4]) : 0) * 262144ll;
  float* buf = nullptr; CK(hipMalloc(&buf, (floats + 1) * 4)); CK(hipMemset(buf, 0, (floats + 1) * 4));
  const int wgs = floats ? 2048 : 64;
  hipStream_t s, s2; CK(hipStreamCreate(&s)); CK(hipStreamCreate(&s2));
  hipEvent_t fork, join; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
  auto chain = [&](bool forked) {
    for (int i = 0; i < N; ++i) {
      hipLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, s, cyc, buf, floats);
      if (forked && i % F == F - 1) {  // a side kernel beside the chain, joined one kernel later
        hipEventRecord(fork, s); hipStreamWaitEvent(s2, fork, 0);
        hipLaunchKernelGGL(spin, dim3(8), dim3(256), 0, s2, cyc, buf + floats, 0ll);
        hipEventRecord(join, s2); hipStreamWaitEvent(s, join, 0);
      }
    }
  };
  int rtv = 0, drv = 0; hipRuntimeGetVersion(&rtv); hipDriverGetVersion(&drv);
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  printf("%s, HIP runtime %d, driver %d; %d dependent kernels of %lld cycles (%d workgroups, %lld MB rewritten per kernel), fork/join every %d kernels in the forked modes\n", pr.name, rtv, drv, N, cyc, wgs, floats / 262144, F);
  for (int forked = 0; forked < 2; ++forked) {
    chain(forked); CK(hipDeviceSynchronize());
    double t = now();
    for (int r = 0; r < reps; ++r) chain(forked);
    CK(hipStreamSynchronize(s)); CK(hipDeviceSynchronize());
    const double eager = (now() - t) / reps / N;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal)); chain(forked); CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    t = now();
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    const double graph = (now() - t) / reps / N;
    printf("%-28s eager %.2f us per kernel, graph replay %.2f us per kernel (replay - eager = %+.2f us per node)\n",
           forked ? "with fork/join side kernels:" : "one linear stream:", eager, graph, graph - eager);
  }
  return 0;
}
