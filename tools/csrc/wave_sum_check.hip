#include <hip/hip_runtime.h>
#include <cstdio>
#include "common.h"
__global__ void k(const float* in, float* out) { float v = in[threadIdx.x]; float s = wave_sum(v); out[threadIdx.x] = s; }
int main() { float h[128], o[128]; double ref[2] = {0, 0}; for (int i = 0; i < 128; ++i) { h[i] = (float)((i * 37) % 101) - 50.f + 0.25f * i; ref[i / 64] += h[i]; }
  float *d, *e; hipMalloc(&d, 512); hipMalloc(&e, 512); hipMemcpy(d, h, 512, hipMemcpyHostToDevice);
  k<<<1, 128>>>(d, e); hipMemcpy(o, e, 512, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 128; ++i) if (o[i] != (float)ref[i / 64]) ++bad;
  printf("wave_sum: lane0 %.3f (ref %.3f) lane100 %.3f (ref %.3f) mismatching lanes %d\n", o[0], ref[0], o[100], ref[1], bad); return bad != 0; }
