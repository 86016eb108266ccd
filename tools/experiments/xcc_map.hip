// Which XCD runs workgroup i of a 1-D / 2-D grid?  Prints the XCC_ID hardware register (s_getreg HW_REG_XCC_ID) per
// workgroup of a 256-workgroup launch, and whether it equals (linear workgroup id % 8) everywhere.
//   hipcc --offload-arch=gfx950 -O2 -o xcc_map xcc_map.hip && ./xcc_map
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void k(unsigned* out) {
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID[3:0]
  if (threadIdx.x == 0) out[blockIdx.y * gridDim.x + blockIdx.x] = xcc;
}

int main() {
  unsigned* d;
  hipMalloc((void**)&d, 4096 * sizeof(unsigned));
  const dim3 grids[3] = {dim3(256, 1), dim3(8, 32), dim3(8, 64)};
  for (const dim3& g : grids) {
    const int n = g.x * g.y;
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(k, g, dim3(256), 0, 0, d);
      hipDeviceSynchronize();
    }
    unsigned h[4096];
    hipMemcpy(h, d, n * sizeof(unsigned), hipMemcpyDeviceToHost);
    int same = 0, cnt[16] = {0};
    for (int i = 0; i < n; ++i) {
      same += (h[i] & 15) == (unsigned)(i % 8);
      ++cnt[h[i] & 15];
    }
    printf("grid (%d, %d): xcc == id %% 8 for %d of %d workgroups; first 24:", g.x, g.y, same, n);
    for (int i = 0; i < 24; ++i) printf(" %u", h[i] & 15);
    printf(" | per xcc:");
    for (int i = 0; i < 8; ++i) printf(" %d", cnt[i]);
    printf("\n");
  }
  return 0;
}
