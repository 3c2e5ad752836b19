// LDS-DMA probe: does an out-of-range lane of `buffer_load ... lds` write zeros or skip?  (it writes zeros)
// hipcc --offload-arch=gfx950 -O3 -x hip tools/csrc/dma_probe.hip -o tools/csrc/dma_probe && tools/csrc/dma_probe
#include <hip/hip_runtime.h>
__global__ void k(const float* g, float* out, int n) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g), 0, n * 4, 0x00020000);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned off = (threadIdx.x * 4) * 4u;
  if (threadIdx.x >= 200) off = 0xfffffff0u;
  for (int i = threadIdx.x; i < 1024; i += 256) sm[i] = -1.f;
  __syncthreads();
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(sm + wave * 256), 16, off, 0, 0, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 256) out[i] = sm[i];
}
int main() {
  float *g, *o; hipMalloc(&g, 4096); hipMalloc(&o, 4096);
  float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = i + 1;
  hipMemcpy(g, h, 4096, hipMemcpyHostToDevice);
  k<<<1, 256, 4096>>>(g, o, 1024);
  hipMemcpy(h, o, 4096, hipMemcpyDeviceToHost);
  printf("in-range [0]=%g [799]=%g | OOB lanes (thread 200..255): [800]=%g [803]=%g [1023]=%g\n", h[0], h[799], h[800], h[803], h[1023]);
  return 0;
}
