// Does a hipGraph run independent branches concurrently on MI355X / ROCm 7.2, and what does a replay cost on the host?
// Captures (stream capture, fork / join through events) 4 branches x 6 dependent kernels of ~25 us each (4 workgroups: they cannot fill
// the chip, so concurrency is visible) and compares: (a) the same 24 launches eagerly on 4 streams, (b) one graph launch per iteration.
// Prints host time per iteration spent in the launch calls and the wall time per iteration.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/graph_branches.hip -o /tmp/graph_branches && /tmp/graph_branches
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(long long cycles, int* sink) {
  const long long t0 = clock64();
  while (clock64() - t0 < cycles) {}
  if (threadIdx.x == 0 && blockIdx.x == 0 && cycles < 0) *sink = 1;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const int NB = 4, NK = 6, ITERS = 300;
  const long long cyc = 50000;   // ~25 us at ~2 GHz
  hipStream_t st[NB];
  for (int b = 0; b < NB; b++) CK(hipStreamCreateWithFlags(&st[b], hipStreamNonBlocking));
  int* sink; CK(hipMalloc(&sink, 4));
  hipEvent_t fork, join[NB];
  CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
  for (int b = 0; b < NB; b++) CK(hipEventCreateWithFlags(&join[b], hipEventDisableTiming));
  // warm
  for (int b = 0; b < NB; b++) hipLaunchKernelGGL(spin, dim3(4), dim3(64), 0, st[b], cyc, sink);
  CK(hipDeviceSynchronize());
  // (a) eager, 4 streams
  double host = 0; double t0 = now();
  for (int it = 0; it < ITERS; it++) {
    const double h0 = now();
    for (int k = 0; k < NK; k++) for (int b = 0; b < NB; b++) hipLaunchKernelGGL(spin, dim3(4), dim3(64), 0, st[b], cyc, sink);
    host += now() - h0;
    if (it % 8 == 7) for (int b = 0; b < NB; b++) CK(hipStreamSynchronize(st[b]));   // keep the queues from running away
  }
  CK(hipDeviceSynchronize());
  double t1 = now();
  printf("eager  4 streams x %d kernels: wall %.1f us / iteration, host in launch calls %.1f us / iteration (%.2f us per launch)\n", NK, 1e6 * (t1 - t0) / ITERS,
         1e6 * host / ITERS, 1e6 * host / ITERS / (NB * NK));
  // (b) graph: capture on st[0], fork to the others
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st[0], hipStreamCaptureModeThreadLocal));
  CK(hipEventRecord(fork, st[0]));
  for (int b = 1; b < NB; b++) CK(hipStreamWaitEvent(st[b], fork, 0));
  for (int k = 0; k < NK; k++) for (int b = 0; b < NB; b++) hipLaunchKernelGGL(spin, dim3(4), dim3(64), 0, st[b], cyc, sink);
  for (int b = 1; b < NB; b++) { CK(hipEventRecord(join[b], st[b])); CK(hipStreamWaitEvent(st[0], join[b], 0)); }
  CK(hipStreamEndCapture(st[0], &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, st[0])); CK(hipStreamSynchronize(st[0]));
  host = 0; t0 = now();
  for (int it = 0; it < ITERS; it++) {
    const double h0 = now();
    CK(hipGraphLaunch(ge, st[0]));
    host += now() - h0;
    if (it % 8 == 7) CK(hipStreamSynchronize(st[0]));
  }
  CK(hipStreamSynchronize(st[0]));
  t1 = now();
  printf("graph  4 branches x %d kernels: wall %.1f us / iteration, host in hipGraphLaunch %.1f us / iteration\n", NK, 1e6 * (t1 - t0) / ITERS, 1e6 * host / ITERS);
  printf("(one branch alone is %d x ~25 us = ~%d us; serialised branches would be ~%d us)\n", NK, NK * 25, NB * NK * 25);
  // (c) two graphs launched alternately on two streams: do consecutive graph launches overlap?
  hipGraphExec_t ge2; CK(hipGraphInstantiate(&ge2, g, nullptr, nullptr, 0));
  t0 = now();
  for (int it = 0; it < ITERS; it++) {
    CK(hipGraphLaunch(it & 1 ? ge2 : ge, st[it & 1]));
    if (it % 8 == 7) { CK(hipStreamSynchronize(st[0])); CK(hipStreamSynchronize(st[1])); }
  }
  CK(hipDeviceSynchronize());
  t1 = now();
  printf("graph  two execs alternating on two streams: wall %.1f us / iteration\n", 1e6 * (t1 - t0) / ITERS);
  return 0;
}
