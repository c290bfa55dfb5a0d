// Which XCD do the blocks of a launch run on? Prints the XCC_ID register value of blocks 0..31 and a histogram over 2048 blocks.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20);
}
int main() {
    unsigned* d; const int n = 2048;
    if (hipMalloc(&d, n * 4) != hipSuccess) return 1;
    k<<<n, 256>>>(d);
    unsigned h[n];
    if (hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    for (int i = 0; i < 32; ++i) printf("%u ", h[i]);
    printf("\n");
    int hist[16] = {0};
    for (int i = 0; i < n; ++i) hist[h[i] & 15]++;
    for (int i = 0; i < 16; ++i) printf("%d:%d ", i, hist[i]);
    printf("\n");
    return 0;
}
