// What a grid-wide barrier costs on this GPU, against the 4.4-6.7 us a short kernel of the grid build takes from start to
// end: a persistent kernel whose blocks (one per CU at most, so all are resident) meet at a counter N times.
//   hipcc --offload-arch=gfx950 -O2 -o grid_barrier tools/experiments/grid_barrier.hip && ./grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_barriers(unsigned int* counter, unsigned int* gave_up, unsigned int* sink, int n_barriers) {
    unsigned int seen = 0;
    for (int b = 1; b <= n_barriers; ++b) {
        // a little work whose result the others read after the barrier, as a build phase would leave
        if (threadIdx.x == 0) sink[blockIdx.x] = (unsigned int)b;
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicAdd(counter, 1u);
            const unsigned int target = (unsigned int)b * gridDim.x;
            unsigned int spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 4000000u) { atomicExch(gave_up, 1u); break; }     // never hang the GPU
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) seen += sink[(blockIdx.x + 1) % gridDim.x];
    }
    if (threadIdx.x == 0 && seen == 0xffffffffu) sink[0] = seen;
}
__global__ void k_empty(unsigned int* sink) { if (threadIdx.x == 0 && blockIdx.x == 0) sink[0] += 1; }

int main() {
    unsigned int *counter, *gave_up, *sink;
    CHECK(hipMalloc(&counter, 4)); CHECK(hipMalloc(&gave_up, 4)); CHECK(hipMalloc(&sink, 4096 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int N = 200;
    for (int blocks : {64, 128, 256}) {
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipMemset(counter, 0, 4)); CHECK(hipMemset(gave_up, 0, 4));
            CHECK(hipEventRecord(e0));
            k_barriers<<<blocks, 256>>>(counter, gave_up, sink, N);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            unsigned int g; CHECK(hipMemcpy(&g, gave_up, 4, hipMemcpyDeviceToHost));
            if (rep) printf("%3d blocks of 256 threads: %.2f us per grid barrier (%d barriers)%s\n", blocks, ms * 1e3 / N, N, g ? "  [a block gave up waiting]" : "");
        }
    }
    // a chain of empty kernels on one stream, for comparison
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < N; ++i) k_empty<<<256, 256>>>(sink);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("chain of %d empty kernels (256 blocks) on one stream: %.2f us per kernel\n", N, ms * 1e3 / N);
    }
    return 0;
}
