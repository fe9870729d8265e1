// Microbenchmark: the same 55-wide f64 atomic add issued as global_atomic_add_f64 (address space known to the
// compiler) and as flat_atomic_add_f64 (pointer loaded from memory, as the kernels get P.scaspe[l]).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(1))) double gdouble;
template <int MODE>
__global__ void k(double* tab, double** ptab, unsigned rows, int rowlen, int width, int iters, unsigned seed)
{
    const int lane = threadIdx.x & 63;
    unsigned s = seed + (blockIdx.x * blockDim.x + threadIdx.x) / 64 * 7919u;
    double* base = (MODE == 0) ? tab : ptab[0];  // MODE 1/2: generic pointer from memory
    for (int i = 0; i < iters; i++) {
        s = s * 1664525u + 1013904223u;
        size_t row = (s >> 8) % rows;
        double* a = base + row * rowlen + lane;
        if (lane < width) {
            if (MODE == 2) __hip_atomic_fetch_add((gdouble*)a, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // cast back to global
            else atomicAdd(a, 1.0);
        }
    }
}
template <int MODE> void run(const char* name, double* d, double** dp, unsigned rows, int blocks = 1024)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4096;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, dp, rows, 55, 55, 16, 1u);
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, dp, rows, 55, 55, iters, 7u); (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s blocks %5d %8.3f ms  %.3e wave-atomics/s\n", name, blocks, ms, (double)blocks * 4 * iters / ms * 1e3);
}
int main()
{
    const unsigned rows = 1u << 21;
    double* d; (void)hipMalloc(&d, (size_t)rows * 55 * 8); (void)hipMemset(d, 0, (size_t)rows * 55 * 8);
    double** dp; (void)hipMalloc(&dp, sizeof(double*)); (void)hipMemcpy(dp, &d, sizeof(double*), hipMemcpyHostToDevice);
    run<0>("global (kernel argument)", d, dp, rows);
    run<1>("flat (pointer from memory)", d, dp, rows);
    run<2>("pointer from memory, cast", d, dp, rows);
    // how the rate depends on the number of workgroups (256 CUs; one 256-thread workgroup = 4 waves)
    for (int b : {32, 64, 128, 256, 512, 2048}) run<0>("global", d, dp, rows, b);
    return 0;
}
