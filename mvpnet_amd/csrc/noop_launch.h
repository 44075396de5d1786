// noop_launch.h -- force-included (-include) by the `noop` target of the Makefile ONLY: the same sources, the same C ABI, the same
// argument checks and host logic, but no kernel is ever launched (and no device memset / copy is queued).  The result,
// mvpnet_amd/libmvp_noop.so, is a measurement aid: `MVP_LIBRARY=.../libmvp_noop.so` makes a process run the real Python training step --
// every ctypes call, every allocation, every autograd node -- without GPU work of ours, so several such "launch-only" peers load the
// host exactly as ranks of a multi-GPU job do while ONE real rank is timed (tools/multi_rank_host.sh).  Results
// of such a process are garbage by construction; nothing in the product or the tests loads this library.
#pragma once
#include <hip/hip_runtime.h>
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(...) ((void)0)
#define hipMemsetAsync(...) (hipSuccess)
#define hipMemcpyAsync(...) (hipSuccess)
