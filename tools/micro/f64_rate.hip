// Microbenchmark: issue cost and dependent-issue latency of the f64 vector instructions the HG weight x**-1.5 is made of
// (v_fma_f64, v_mul_f64, v_rsq_f64, v_cndmask pairs), one wave per SIMD and three (the stage kernel's occupancy).
// A "chain" is N instructions each depending on the previous one; "ilpK" interleaves K independent chains.
// Cycles are s_memtime ticks of one wave divided by its instruction count; with W waves per SIMD the SIMD-side cost per
// instruction is that number / W when the waves overlap perfectly.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/f64_rate.hip -o tools/micro/f64_rate.bin && tools/micro/f64_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int KIND, int ILP>
__global__ void k(double* out, int iters, unsigned long long* cyc)
{
    double a0 = 1.0 + threadIdx.x * 1e-9, a1 = 1.1 + threadIdx.x * 1e-9, a2 = 1.2, a3 = 1.3;
    const double b = 0.9999999, c = 1e-9;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {  // v_fma_f64
            if (ILP == 1) { REP16(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));) }
            if (ILP == 2) { REP16(asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(a0), "+v"(a1) : "v"(b), "v"(c));) }
            if (ILP == 4) { REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        } else if (KIND == 1) {  // v_mul_f64
            if (ILP == 1) { REP16(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a0) : "v"(b));) }
            if (ILP == 2) { REP16(asm volatile("v_mul_f64 %0, %0, %2\n v_mul_f64 %1, %1, %2" : "+v"(a0), "+v"(a1) : "v"(b));) }
            if (ILP == 4) { REP16(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        } else if (KIND == 2) {  // v_rsq_f64 (x -> rsq(x): converges to 1, stays finite)
            if (ILP == 1) { REP16(asm volatile("v_rsq_f64 %0, %0" : "+v"(a0));) }
            if (ILP == 2) { REP16(asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1" : "+v"(a0), "+v"(a1));) }
            if (ILP == 4) { REP16(asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        } else if (KIND == 3) {  // v_rsq_f32
            float f0 = (float)a0, f1 = (float)a1, f2 = (float)a2, f3 = (float)a3;
            if (ILP == 1) { REP16(asm volatile("v_rsq_f32 %0, %0" : "+v"(f0));) }
            if (ILP == 2) { REP16(asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1" : "+v"(f0), "+v"(f1));) }
            if (ILP == 4) { REP16(asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));) }
            a0 = f0; a1 = f1; a2 = f2; a3 = f3;
        } else if (KIND == 4) {  // the whole weight: x -> x**-1.5 as pow_m1p5_f writes it (fma, rsq, 5 mul/fma) + accumulate, per chain
            double x0 = a0, x1 = a1, x2 = a2, x3 = a3;
#define POW(x, acc) { double y = __builtin_amdgcn_rsq(x); const double hx = 0.5 * x; y = y * __builtin_fma(-(hx * y), y, 1.5); acc = __builtin_fma(c, (y * y) * y, acc); }
            if (ILP == 1) { for (int r = 0; r < 16; r++) { POW(x0, a0); x0 = a0 + 1.0; } }
            if (ILP == 2) { for (int r = 0; r < 16; r++) { POW(x0, a0); POW(x1, a1); x0 = a0 + 1.0; x1 = a1 + 1.0; } }
            if (ILP == 4) { for (int r = 0; r < 16; r++) { POW(x0, a0); POW(x1, a1); POW(x2, a2); POW(x3, a3); x0 = a0 + 1.0; x1 = a1 + 1.0; x2 = a2 + 1.0; x3 = a3 + 1.0; } }
        } else if (KIND == 5) {  // v_cvt_f32_f64 + v_rsq_f32 + v_cvt_f64_f32: a single-precision seed
            if (ILP == 1) { REP16(asm volatile("v_cvt_f32_f64 %0, %1\n v_rsq_f32 %0, %0\n v_cvt_f64_f32 %1, %0" : "+v"(*(float*)&a1), "+v"(a0));) }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND, int ILP> void run(const char* name, int waves_per_simd, int per_iter)
{
    double* d; unsigned long long* c;
    const int blocks = 256, threads = 256 * waves_per_simd;  // one block per CU, waves_per_simd waves on each of its 4 SIMDs
    hipMalloc(&d, (size_t)blocks * threads * sizeof(double)); hipMalloc(&c, 8);
    const int iters = 2000;
    hipLaunchKernelGGL((k<KIND, ILP>), dim3(blocks), dim3(threads), 0, 0, d, 10, c);
    hipLaunchKernelGGL((k<KIND, ILP>), dim3(blocks), dim3(threads), 0, 0, d, iters, c);
    hipDeviceSynchronize();
    unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * per_iter;
    printf("%-34s ilp %d, %d wave(s)/SIMD: %7.2f ticks per instruction of one wave, %7.2f per instruction per SIMD\n", name, ILP, waves_per_simd, h / n, h / n / waves_per_simd);
    hipFree(d); hipFree(c);
}

int main()
{
    for (int w : {1, 3}) {
        run<0, 1>("v_fma_f64", w, 16); run<0, 2>("v_fma_f64", w, 32); run<0, 4>("v_fma_f64", w, 64);
        run<1, 1>("v_mul_f64", w, 16); run<1, 2>("v_mul_f64", w, 32); run<1, 4>("v_mul_f64", w, 64);
        run<2, 1>("v_rsq_f64", w, 16); run<2, 2>("v_rsq_f64", w, 32); run<2, 4>("v_rsq_f64", w, 64);
        run<3, 1>("v_rsq_f32", w, 16); run<3, 2>("v_rsq_f32", w, 32); run<3, 4>("v_rsq_f32", w, 64);
        run<4, 1>("x**-1.5 + fma (9 instr)", w, 16); run<4, 2>("x**-1.5 + fma (9 instr)", w, 32); run<4, 4>("x**-1.5 + fma (9 instr)", w, 64);
        run<5, 1>("cvt + rsq_f32 + cvt (3 instr)", w, 16);
    }
    {   // the tick of s_memtime against the wall clock: a known number of fma at a known rate is not assumed; time a long run
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        double* d; unsigned long long* c; hipMalloc(&d, 256 * 256 * 8); hipMalloc(&c, 8);
        hipEventRecord(e0); hipLaunchKernelGGL((k<0, 1>), dim3(256), dim3(256), 0, 0, d, 200000, c); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        printf("s_memtime: %.1f ticks per microsecond (%.3f ms for %llu ticks)\n", h / (ms * 1e3), ms, h);
    }
    return 0;
}
