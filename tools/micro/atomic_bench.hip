// Microbenchmark: throughput of no-return f64 / f32 global atomic adds shaped like the scaspe axpy
// (one wave instruction = `width` contiguous elements of one random row of a [rows][rowlen] table).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <typename T>
__global__ void k(T* tab, int rows, int rowlen, int width, int iters, unsigned seed, int scattered)
{
    const int lane = threadIdx.x & 63;
    unsigned s = seed + (blockIdx.x * blockDim.x + threadIdx.x) / 64 * 7919u;
    for (int i = 0; i < iters; i++) {
        s = s * 1664525u + 1013904223u;
        unsigned row = (s >> 8) % (unsigned)rows;
        if (scattered) { unsigned s2 = s * 2654435761u + lane * 40503u; row = (s2 >> 8) % (unsigned)rows; }
        if (lane < width) atomicAdd(&tab[(size_t)row * rowlen + (scattered ? 0 : lane)], (T)1.0);
    }
}
template <typename T> void run(const char* name, int rows, int rowlen, int width, int scattered)
{
    T* d; hipMalloc(&d, (size_t)rows * rowlen * sizeof(T)); hipMemset(d, 0, (size_t)rows * rowlen * sizeof(T));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int blocks = 256 * 4, iters = 2000;
    hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256), 0, 0, d, rows, rowlen, width, 10, 1u, scattered);
    hipEventRecord(e0); hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256), 0, 0, d, rows, rowlen, width, iters, 7u, scattered); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    double ninstr = (double)blocks * 4 * iters;
    printf("%-10s rows %8d rowlen %3d width %2d %s: %8.3f ms  %.3e wave-atomics/s  %.1f GB/s added  %.1f ns per instr per CU\n", name, rows, rowlen, width,
           scattered ? "scattered" : "contiguous", ms, ninstr / ms * 1e3, ninstr * width * sizeof(T) / ms * 1e3 / 1e9, ms * 1e6 / (ninstr / 256));
    hipFree(d);
}
int main()
{
    run<float>("f32", 1 << 20, 64, 64, 0);
    run<double>("f64", 1 << 20, 64, 64, 0);
    run<double>("f64", 1 << 20, 55, 55, 0);
    run<double>("f64", 1 << 20, 56, 56, 0);
    run<double>("f64", 1 << 14, 55, 55, 0);
    run<double>("f64", 1 << 20, 55, 32, 0);
    run<double>("f64", 1 << 20, 55, 16, 0);
    run<double>("f64", 1 << 20, 55, 8, 0);
    run<double>("f64", 1 << 22, 1, 64, 1);
    run<float>("f32", 1 << 22, 1, 64, 1);
    return 0;
}
