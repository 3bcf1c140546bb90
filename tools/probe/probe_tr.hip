// What does ds_read_b64_tr_b16 return?  LDS element e holds the value e; lane l passes the address of element addr_of(l).
//   hipcc --offload-arch=gfx950 -O2 probe_tr.hip -o probe_tr && ./probe_tr        (GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(const int* addr, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
  for (int e = threadIdx.x; e < 8192; e += 64) lds[e] = (unsigned short)e;
  __syncthreads();
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + addr[threadIdx.x]));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
int main() {
  int h_addr[64]; unsigned short h_out[256];
  int* d_addr; unsigned short* d_out;
  hipMalloc(&d_addr, sizeof h_addr); hipMalloc(&d_out, sizeof h_out);
  for (int mode = 0; mode < 2; ++mode) {
    // mode 0: lane-linear (lane l -> element 4 l).  mode 1: row-major [k][128] image: lane i of a 16-lane group -> row (i >> 2), column 4 (i & 3) + 16 * group
    for (int l = 0; l < 64; ++l) h_addr[l] = mode == 0 ? 4 * l : ((l & 15) >> 2) * 128 + 4 * (l & 3) + 16 * (l >> 4);
    hipMemcpy(d_addr, h_addr, sizeof h_addr, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d:", l);
      for (int j = 0; j < 4; ++j) {
        printf(" %5d", h_out[4 * l + j]);
        // expectation: lane i of group g, element j = piece of lane (4 j + i / 4) of the same group, its element i % 4
        const int i = l & 15, g = l >> 4, src = 16 * g + 4 * j + (i >> 2);
        if (h_out[4 * l + j] != h_addr[src] + (i & 3)) ++bad;
      }
      printf("\n");
    }
    printf("mode %d: %d values differ from the expected transpose\n", mode, bad);
  }
  return 0;
}
