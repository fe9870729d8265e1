// Microbenchmark: f64 no-return atomic adds of `width` contiguous elements, `burst` consecutive rows per
// random pick (models the scaspe deposit of one ray-cell intersection for `burst` wavelengths when the
// layout is cell-major), versus burst = 1 (wavelength-major layout: every row is a separate random place).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k(double* tab, unsigned picks, int rowlen, int width, int burst, int iters, unsigned seed)
{
    const int lane = threadIdx.x & 63;
    unsigned s = seed + (blockIdx.x * blockDim.x + threadIdx.x) / 64 * 7919u;
    for (int i = 0; i < iters; i++) {
        s = s * 1664525u + 1013904223u;
        size_t row0 = (size_t)((s >> 8) % picks) * burst;
        for (int b = 0; b < burst; b++)
            if (lane < width) atomicAdd(&tab[(row0 + b) * rowlen + lane], 1.0);
    }
}
static void run(size_t rows, int rowlen, int width, int burst)
{
    double* d; hipMalloc(&d, rows * rowlen * sizeof(double)); hipMemset(d, 0, rows * rowlen * sizeof(double));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int blocks = 256 * 4, iters = 4096 / burst;
    unsigned picks = (unsigned)(rows / burst);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, picks, rowlen, width, burst, 4, 1u);
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, picks, rowlen, width, burst, iters, 7u); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    double ninstr = (double)blocks * 4 * iters * burst;
    printf("table %7.1f MB rowlen %3d width %2d burst %3d: %8.3f ms  %.3e wave-atomics/s  %.1f GB/s added\n", rows * rowlen * 8 / 1e6, rowlen, width,
           burst, ms, ninstr / ms * 1e3, ninstr * width * 8 / ms * 1e3 / 1e9);
    hipFree(d);
}
int main()
{
    for (size_t rows : {(size_t)1 << 14, (size_t)1 << 17, (size_t)1 << 20, (size_t)1 << 22})
        for (int burst : {1, 4, 32}) run(rows, 55, 55, burst);
    for (int burst : {1, 32}) run((size_t)1 << 22, 56, 55, burst);
    for (int burst : {1, 32}) run((size_t)1 << 22, 64, 55, burst);
    for (int burst : {1, 32}) run((size_t)1 << 22, 64, 64, burst);
    return 0;
}
