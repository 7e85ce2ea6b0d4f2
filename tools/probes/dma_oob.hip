// Probe: does an out-of-range lane of `buffer_load_dword ... lds` write 0 into LDS, or skip the write?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* src, float* dst, int nbytes) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 512; i += 256) lds[i] = -7.f;       // poison
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  int voff = (lane % 3 == 0) ? -1 : lane * 4;                        // every third lane out of range
  if (lane < 48)                                                     // lanes 48..63 masked by EXEC
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + wave * 64), 4, voff, wave * 256, 0, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 256) dst[i] = lds[i];
}
int main() {
  std::vector<float> h(1024); for (int i = 0; i < 1024; ++i) h[i] = 100.f + i;
  float *s, *d; hipMalloc(&s, 4096); hipMalloc(&d, 2048); hipMemcpy(s, h.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, s, d, 4096);
  std::vector<float> o(512); hipMemcpy(o.data(), d, 2048, hipMemcpyDeviceToHost);
  for (int w = 0; w < 2; ++w) { printf("wave %d:", w); for (int l = 0; l < 64; ++l) printf(" %g", o[w * 64 + l]); printf("\n"); }
  return 0;
}
