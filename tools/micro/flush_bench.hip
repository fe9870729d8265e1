// Microbenchmark of the scattering-deposit inner loop (process_scatt_rad restated for a wave):
// every wave owns NREC synthetic records {cell, direction, C[4], U[4]}, sorted so that runs of RUN
// consecutive records share a cell, and adds sum_m C_m * (ga - gb cos)^-1.5 to the 55 slots of the
// cell's row for 4 wavelengths. Variants isolate what the loop costs:
//   0  readlane broadcast of every record field (the production loop)
//   1  same without the atomics
//   2  records staged in LDS, fields fetched with broadcast ds_read
//   3  like 0 without the per-wavelength zero test
//   4  two records per iteration (independent weight chains overlap)
//   5  one basic block for records whose four wavelengths are all live
//   6  variant 4 without the atomics
// usage: flush_bench.bin [run_length] [repetitions]
// MI355X: run length 7: 3.9e9 records/s (4.9e9 without atomics); run length 3: 2.0e9 (4.3e9 without): with few
// records per cell the loop is bound by the 2.96e9 f64 atomic instructions/s of the chip, and none of the variants
// 2-5 changes that.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

#define LB 4
#define NP 55
__device__ __forceinline__ double readlane_d(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double pow_m1p5(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    y = y * __builtin_fma(-((0.5 * x) * y), y, 1.5);
    return y * y * y;
}
struct Rec { const int* cell; const double *nx, *ny, *nz, *C, *U; int cap; };

template <int VAR>
__global__ void __launch_bounds__(256, 2) k(Rec R, int nrec, double* scaspe, double* u_final, int ncell, const double* tab, double g0, int reps)
{
    __shared__ double lds[4][64 * 12];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int gwave = blockIdx.x * 4 + wib;
    const long long base0 = (long long)gwave * nrec;
    const double tx = lane < NP ? tab[lane] : 0, ty = lane < NP ? tab[NP + lane] : 0, tz = lane < NP ? tab[2 * NP + lane] : 0;
    double gak[LB], gbk[LB];
#pragma unroll
    for (int k2 = 0; k2 < LB; k2++) { const double g = g0 + 0.05 * k2; gak[k2] = 1.0 + g * g; gbk[k2] = 2.0 * g; }
    for (int rep = 0; rep < reps; rep++)
    for (int b = 0; b < nrec; b += 64) {
        const long long i = base0 + b + lane;
        const int key = R.cell[i];
        const double hx = R.nx[i], hy = R.ny[i], hz = R.nz[i];
        double Cv[LB], Uv[LB];
#pragma unroll
        for (int k2 = 0; k2 < LB; k2++) { Cv[k2] = R.C[(long long)k2 * R.cap + i]; Uv[k2] = R.U[(long long)k2 * R.cap + i]; }
        {
            double chk = hx + hy + hz;
#pragma unroll
            for (int k2 = 0; k2 < LB; k2++) chk += Cv[k2] + Uv[k2];
            asm volatile("" ::"v"(chk), "v"(key));
        }
        if (VAR == 2) {
            double* L = lds[wib];
            L[lane * 12 + 0] = hx; L[lane * 12 + 1] = hy; L[lane * 12 + 2] = hz;
#pragma unroll
            for (int k2 = 0; k2 < LB; k2++) { L[lane * 12 + 3 + k2] = Cv[k2]; L[lane * 12 + 7 + k2] = Uv[k2]; }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        }
        unsigned long long pend = ~0ull;
        while (pend) {
            const int s = __ffsll((long long)pend) - 1;
            const int kk = __builtin_amdgcn_readlane(key, s);
            unsigned long long grp = __ballot(key == kk);
            pend &= ~grp;
            double acc[LB], su[LB];
#pragma unroll
            for (int k2 = 0; k2 < LB; k2++) { acc[k2] = 0.0; su[k2] = 0.0; }
            if (VAR == 4 || VAR == 6) {
                // two members per iteration: their weight chains are independent and overlap
                while (grp) {
                    const int r0 = __ffsll((long long)grp) - 1;
                    grp &= grp - 1;
                    const bool two = grp != 0;
                    const int r1 = two ? __ffsll((long long)grp) - 1 : r0;
                    if (two) grp &= grp - 1;
                    const double ax = readlane_d(hx, r0), ay = readlane_d(hy, r0), az = readlane_d(hz, r0);
                    const double bx = readlane_d(hx, r1), by = readlane_d(hy, r1), bz = readlane_d(hz, r1);
                    const double ca = tx * ax + ty * ay + tz * az, cb = tx * bx + ty * by + tz * bz;
#pragma unroll
                    for (int k2 = 0; k2 < LB; k2++) {
                        const double Ca = readlane_d(Cv[k2], r0), Cb = two ? readlane_d(Cv[k2], r1) : 0.0;
                        su[k2] += readlane_d(Uv[k2], r0) + (two ? readlane_d(Uv[k2], r1) : 0.0);
                        if (Ca != 0.0 || Cb != 0.0) {
                            const double wa = pow_m1p5(__builtin_fma(-gbk[k2], ca, gak[k2])), wb = pow_m1p5(__builtin_fma(-gbk[k2], cb, gak[k2]));
                            acc[k2] = __builtin_fma(Ca, wa, __builtin_fma(Cb, wb, acc[k2]));
                        }
                    }
                }
            } else
            while (grp) {
                const int r = __ffsll((long long)grp) - 1;
                grp &= grp - 1;
                double nx, ny, nz, Cm[LB], Um[LB];
                if (VAR == 2) {
                    const double* L = lds[wib] + r * 12;
                    nx = L[0]; ny = L[1]; nz = L[2];
#pragma unroll
                    for (int k2 = 0; k2 < LB; k2++) { Cm[k2] = L[3 + k2]; Um[k2] = L[7 + k2]; }
                } else {
                    nx = readlane_d(hx, r); ny = readlane_d(hy, r); nz = readlane_d(hz, r);
#pragma unroll
                    for (int k2 = 0; k2 < LB; k2++) { Cm[k2] = readlane_d(Cv[k2], r); Um[k2] = readlane_d(Uv[k2], r); }
                }
                const double cs0 = tx * nx + ty * ny + tz * nz;
                if (VAR == 5) {
                    // all four wavelengths live: one basic block, four independent chains
                    const bool all4 = Cm[0] != 0.0 && Cm[1] != 0.0 && Cm[2] != 0.0 && Cm[3] != 0.0;
                    if (all4) {
                        double w[LB];
#pragma unroll
                        for (int k2 = 0; k2 < LB; k2++) w[k2] = pow_m1p5(__builtin_fma(-gbk[k2], cs0, gak[k2]));
#pragma unroll
                        for (int k2 = 0; k2 < LB; k2++) { su[k2] += Um[k2]; acc[k2] = __builtin_fma(Cm[k2], w[k2], acc[k2]); }
                        continue;
                    }
                }
#pragma unroll
                for (int k2 = 0; k2 < LB; k2++) {
                    su[k2] += Um[k2];
                    if (VAR == 3 || VAR == 2 || Cm[k2] != 0.0) acc[k2] = __builtin_fma(Cm[k2], pow_m1p5(__builtin_fma(-gbk[k2], cs0, gak[k2])), acc[k2]);
                }
            }
#pragma unroll
            for (int k2 = 0; k2 < LB; k2++) {
                double* addr = (lane == 63) ? (u_final + (long long)kk * LB + k2) : (scaspe + ((long long)k2 * ncell + kk) * NP + lane);
                const double v = (lane == 63) ? su[k2] : acc[k2];
                if ((lane < NP || lane == 63) && v != 0.0) {
                    if (VAR == 1 || VAR == 6) asm volatile("" ::"v"(addr), "v"(v));
                    else atomicAdd(addr, v);
                }
            }
        }
    }
}

int main(int argc, char** argv)
{
    const int RUN = argc > 1 ? atoi(argv[1]) : 7;
    const int REPS = argc > 2 ? atoi(argv[2]) : 1;
    const int waves = 2048, nrec = 4096, ncell = 85801;
    const long long N = (long long)waves * nrec;
    std::vector<int> cell(N);
    std::vector<double> nx(N), ny(N), nz(N), C(4 * N), U(4 * N), tab(3 * NP);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) / 16777216.0; };
    int cur = 0;
    for (long long i = 0; i < N; i++) {
        if (i % RUN == 0) cur = (int)(rnd() * ncell) % ncell;
        cell[i] = cur;
        double z = 2 * rnd() - 1, ph = 6.283185307 * rnd(), st = sqrt(1 - z * z);
        nx[i] = st * cos(ph); ny[i] = st * sin(ph); nz[i] = z;
        for (int k2 = 0; k2 < 4; k2++) { C[k2 * N + i] = (rnd() < 0.55) ? rnd() : 0.0; U[k2 * N + i] = rnd(); }
    }
    for (int p = 0; p < NP; p++) { double z = 2 * rnd() - 1, ph = 6.283185307 * rnd(), st = sqrt(1 - z * z); tab[p] = st * cos(ph); tab[NP + p] = st * sin(ph); tab[2 * NP + p] = z; }
    int* dcell; double *dnx, *dny, *dnz, *dC, *dU, *dsc, *du, *dtab;
    hipMalloc(&dcell, N * 4); hipMalloc(&dnx, N * 8); hipMalloc(&dny, N * 8); hipMalloc(&dnz, N * 8);
    hipMalloc(&dC, 4 * N * 8); hipMalloc(&dU, 4 * N * 8); hipMalloc(&dsc, 4ll * ncell * NP * 8); hipMalloc(&du, 4ll * ncell * 8); hipMalloc(&dtab, 3 * NP * 8);
    hipMemcpy(dcell, cell.data(), N * 4, hipMemcpyHostToDevice); hipMemcpy(dnx, nx.data(), N * 8, hipMemcpyHostToDevice);
    hipMemcpy(dny, ny.data(), N * 8, hipMemcpyHostToDevice); hipMemcpy(dnz, nz.data(), N * 8, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), 4 * N * 8, hipMemcpyHostToDevice); hipMemcpy(dU, U.data(), 4 * N * 8, hipMemcpyHostToDevice);
    hipMemcpy(dtab, tab.data(), 3 * NP * 8, hipMemcpyHostToDevice);
    hipMemset(dsc, 0, 4ll * ncell * NP * 8); hipMemset(du, 0, 4ll * ncell * 8);
    Rec R{dcell, dnx, dny, dnz, dC, dU, (int)N};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](int var) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            switch (var) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(waves / 4), dim3(256), 0, 0, R, nrec, dsc, du, ncell, dtab, 0.5, REPS); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(waves / 4), dim3(256), 0, 0, R, nrec, dsc, du, ncell, dtab, 0.5, REPS); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(waves / 4), dim3(256), 0, 0, R, nrec, dsc, du, ncell, dtab, 0.5, REPS); break;
            case 3: hipLaunchKernelGGL(k<3>, dim3(waves / 4), dim3(256), 0, 0, R, nrec, dsc, du, ncell, dtab, 0.5, REPS); break;
            case 4: hipLaunchKernelGGL(k<4>, dim3(waves / 4), dim3(256), 0, 0, R, nrec, dsc, du, ncell, dtab, 0.5, REPS); break;
            case 5: hipLaunchKernelGGL(k<5>, dim3(waves / 4), dim3(256), 0, 0, R, nrec, dsc, du, ncell, dtab, 0.5, REPS); break;
            default: hipLaunchKernelGGL(k<6>, dim3(waves / 4), dim3(256), 0, 0, R, nrec, dsc, du, ncell, dtab, 0.5, REPS); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("variant %d run %d: %8.3f ms  %.3e records/s  %.1f ns per record per wave  (%.0f cycles @2.4GHz)\n", var, RUN, best, (double)N * REPS / best * 1e3,
               best * 1e6 / nrec / REPS, best * 1e6 / nrec / REPS * 2.4);
    };
    for (int v = 0; v < 7; v++) run(v);
    return 0;
}
