// What an event record between two kernels of one stream costs that stream (diagnostic): kernel A
// and kernel B spin for a fixed time and stamp s_memrealtime (100 MHz) at start / end; the gap
// B.start - A.end is printed for (i) nothing in between, (ii) hipEventRecord + a second stream
// waiting on it, (iii) the event attached to A's own dispatch (hipExtLaunchKernelGGL stopEvent).
// hipcc --offload-arch=gfx950 -O3 -x hip tools/csrc/event_gap.hip -o /tmp/event_gap && /tmp/event_gap
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

__global__ void spin_kernel(unsigned long long* stamp, int slot, int ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) stamp[2 * slot] = t0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && blockIdx.x == 0) stamp[2 * slot + 1] = __builtin_amdgcn_s_memrealtime();
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  unsigned long long* stamp;
  CK(hipHostMalloc(&stamp, 64 * sizeof(unsigned long long), hipHostMallocMapped));
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  hipEvent_t ev;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  const int ticks = 10000;   // 100 us
  const dim3 grid(256), block(256);
  for (int mode = 0; mode < 4; ++mode) {
    std::vector<double> gaps, sides;
    for (int rep = 0; rep < 30; ++rep) {
      // a leading kernel keeps the queue busy while the host enqueues the rest
      hipLaunchKernelGGL(spin_kernel, grid, block, 0, s0, stamp, 3, ticks * 3);
      if (mode == 2) {
        hipExtLaunchKernelGGL(spin_kernel, grid, block, 0, s0, nullptr, ev, 0, stamp, 0, ticks);
      } else {
        hipLaunchKernelGGL(spin_kernel, grid, block, 0, s0, stamp, 0, ticks);
        if (mode == 1 || mode == 3) CK(hipEventRecord(ev, s0));
      }
      if (mode >= 1) {
        CK(hipStreamWaitEvent(s1, ev, 0));
        hipLaunchKernelGGL(spin_kernel, dim3(8), block, 0, s1, stamp, 2, ticks / 10);
      }
      if (mode == 3) {   // a second record (what a join costs): stream 0 waits for stream 1's event
        CK(hipEventRecord(ev, s1));
        CK(hipStreamWaitEvent(s0, ev, 0));
      }
      hipLaunchKernelGGL(spin_kernel, grid, block, 0, s0, stamp, 1, ticks);
      CK(hipDeviceSynchronize());
      gaps.push_back(((double)stamp[2] - (double)stamp[1]) / 100.0);
      if (mode >= 1) sides.push_back(((double)stamp[4] - (double)stamp[1]) / 100.0);
    }
    std::sort(gaps.begin(), gaps.end());
    std::sort(sides.begin(), sides.end());
    const char* names[] = {"nothing between A and B", "hipEventRecord after A (+ side stream waiting)",
                           "event attached to A's dispatch (hipExtLaunchKernelGGL stopEvent)",
                           "record + side kernel + main stream waits for the side stream (fork and join)"};
    printf("%-82s gap A.end -> B.start: median %.2f us (min %.2f, max %.2f)", names[mode],
           gaps[gaps.size() / 2], gaps.front(), gaps.back());
    if (mode >= 1) printf("; side kernel starts %.2f us after A.end", sides[sides.size() / 2]);
    printf("\n");
  }
  return 0;
}
