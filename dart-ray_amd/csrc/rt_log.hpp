// rt_log.hpp - the reduce side of the deposit log (rt_wide.hpp, MODE 2): process_scatt_rad (rt_routines.f90:2638-2726)
// as "store every contribution once, then sum per destination" instead of a scatter of atomics. Opt-in (DRT_LOG=1).
//
// rt_stage_wide<LP, 2> leaves one record per (ray, cell) that scattered: {key: cell | HEALPix level << 27 | pixel << 32,
// mask of wavelengths, C = en_sca / HG norm per wavelength of the mask}, one index word per record, and the number of
// records per cell. Here:
//   k_log_scan_*    exclusive scan of the records per cell
//   k_log_scatter   index words -> record offsets grouped by cell
//   k_log_reduce16  per (cell, group of wavelengths): the ray directions are rebuilt from their pixels (the same host-built
//                   tables, hence the same doubles as in the tracing kernel), the Henyey-Greenstein weights of every record
//                   are formed as flush_lambda forms them (same cosine, same x = 1 + g^2 - 2 g cos in fma form; x**-1.5 by a
//                   first-order correction of v_rsq_f64's cube instead of the Newton step, 2e-13 apart) and the rows of the
//                   cell are added to scaspe with plain read-modify-writes: the work item owns them.
// Sums are taken in another order than the atomics of MODE 0 would take them, the weights agree to 2e-13; nothing else differs.
#pragma once
#include "rt_wide.hpp"

#define DRT_LOG_CELL_MASK ((1ull << 27) - 1ull)

// exclusive scan of cnt[0..n) into start[0..n], in three small launches (n up to 1e8 cells): per-block sums, a one-block
// scan of those, per-block exclusive scans with their offsets
#define DRT_SCAN_BLOCK 1024
__global__ void k_log_scan_sums(const unsigned* __restrict__ cnt, int n, unsigned long long* __restrict__ bsum)
{
    __shared__ unsigned long long sh[256];
    unsigned long long v = 0;
    const long long base = (long long)blockIdx.x * DRT_SCAN_BLOCK;
    for (int i = threadIdx.x; i < DRT_SCAN_BLOCK; i += 256) { const long long j = base + i; if (j < n) v += cnt[j]; }
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) bsum[blockIdx.x] = sh[0];
}
__global__ void k_log_scan_top(unsigned long long* __restrict__ bsum, int nb)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        unsigned long long run = 0;
        for (int i = 0; i < nb; i++) { const unsigned long long v = bsum[i]; bsum[i] = run; run += v; }
        bsum[nb] = run;
    }
}
__global__ void k_log_scan_fin(const unsigned* __restrict__ cnt, int n, const unsigned long long* __restrict__ bsum, unsigned long long* __restrict__ start, int nb)
{
    // one thread walks its block's 1024 counts (the scan is a sliver of the pass)
    if (threadIdx.x == 0) {
        unsigned long long run = bsum[blockIdx.x];
        const long long base = (long long)blockIdx.x * DRT_SCAN_BLOCK;
        for (int i = 0; i < DRT_SCAN_BLOCK; i++) { const long long j = base + i; if (j >= n) break; start[j] = run; run += cnt[j]; }
        if ((int)blockIdx.x == nb - 1) start[n] = bsum[nb];
    }
}

__global__ void k_log_scatter(const unsigned long long* __restrict__ idx, const unsigned* __restrict__ chunk_count, unsigned chunk_idx,
                              const unsigned long long* __restrict__ start, unsigned* __restrict__ cursor, unsigned long long* __restrict__ sorted)
{
    const unsigned k = blockIdx.x;   // chunk
    const unsigned n = chunk_count[k];
    for (unsigned e = blockIdx.y * blockDim.x + threadIdx.x; e < n; e += gridDim.y * blockDim.x) {
        const unsigned long long w = idx[(unsigned long long)k * chunk_idx + e];
        const unsigned cell = (unsigned)(w & DRT_LOG_CELL_MASK);
        const unsigned pos = atomicAdd(&cursor[cell], 1u);
        sorted[start[cell] + pos] = w >> 27;  // byte offset / 8 of the record in the log
    }
}

// unit vector of pixel `ipix` at nside = 2^nsl: pix2ang_nest_sc's table rows, hence the doubles the tracing kernel had
__device__ __forceinline__ void log_dir(const DevProblem& P, int nsl, int ipix, double& nx, double& ny, double& nz)
{
    const int jrll[12] = {2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4};
    const int jpll[12] = {1, 3, 5, 7, 0, 2, 4, 6, 1, 3, 5, 7};
    const int nside = 1 << nsl, nl4 = 4 * nside;
    const int face_num = ipix >> (2 * nsl);
    const int ipf = ipix & ((1 << (2 * nsl)) - 1);
    const int ix = (int)compact_even_bits((uint32_t)ipf), iy = (int)compact_even_bits((uint32_t)ipf >> 1);
    const int jrt = ix + iy, jpt = ix - iy;
    const int jr = jrll[face_num] * nside - jrt - 1;
    int nr, kshift;
    if (jr < nside) { nr = jr; kshift = 0; }
    else if (jr <= 3 * nside) { nr = nside; kshift = (jr - nside) & 1; }
    else { nr = nl4 - jr; kshift = 0; }
    int jp = (jpll[face_num] * nr + jpt + 1 + kshift) / 2;
    if (jp > nl4) jp -= nl4;
    if (jp < 1) jp += nl4;
    const double* rt = P.ring_tab[nsl] + 3 * jr;
    const double st = rt[1], ct = rt[2];
    const double* pt = (kshift == 0) ? (P.phi0_tab[nsl] + 3ull * (2ull * nr * (nr - 1) + (jp - 1))) : (P.phi1_tab[nsl] + 3ull * (jp - 1));
    const double sp = pt[1], cp = pt[2];
    nx = st * cp; ny = st * sp; nz = ct;  // calc_cproj (rt_routines.f90:1945), as trace_items_w forms d0, d1, d2
}
// the same with the products left to the caller (nx = st * cp, ny = st * sp, nz = ct): the four table values can be in flight
// while something else is computed
__device__ __forceinline__ void log_dir_loads(const DevProblem& P, int nsl, int ipix, double& st, double& ct, double& sp, double& cp)
{
    const int jrll[12] = {2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4};
    const int jpll[12] = {1, 3, 5, 7, 0, 2, 4, 6, 1, 3, 5, 7};
    const int nside = 1 << nsl, nl4 = 4 * nside;
    const int face_num = ipix >> (2 * nsl);
    const int ipf = ipix & ((1 << (2 * nsl)) - 1);
    const int ix = (int)compact_even_bits((uint32_t)ipf), iy = (int)compact_even_bits((uint32_t)ipf >> 1);
    const int jrt = ix + iy, jpt = ix - iy;
    const int jr = jrll[face_num] * nside - jrt - 1;
    int nr, kshift;
    if (jr < nside) { nr = jr; kshift = 0; }
    else if (jr <= 3 * nside) { nr = nside; kshift = (jr - nside) & 1; }
    else { nr = nl4 - jr; kshift = 0; }
    int jp = (jpll[face_num] * nr + jpt + 1 + kshift) / 2;
    if (jp > nl4) jp -= nl4;
    if (jp < 1) jp += nl4;
    const double* rt = P.ring_tab[nsl] + 3 * jr;
    st = rt[1]; ct = rt[2];
    const double* pt = (kshift == 0) ? (P.phi0_tab[nsl] + 3ull * (2ull * nr * (nr - 1) + (jp - 1))) : (P.phi1_tab[nsl] + 3ull * (jp - 1));
    sp = pt[1]; cp = pt[2];
}

// ---- k_log_reduce16: lanes laid out as 16 slots x 4 records --------------------------------------------------------------------
// A first version made every lane a slot and took the records one after the other: a record's values had to be handed round
// with readlane, a wavelength's constants too, and the scalar side of that (bit counts, selects) cost more than the weights
// themselves (bench step: 2.1 s, of which x**-1.5 0.8 s; this one: 1.6 s). Here a lane is slot (lane & 15) of a block of 16 slots AND record residue (lane >> 4): it takes the records
// 4j + (lane >> 4), j = 0..3, of a sub-tile of 16 records - four independent weight chains per wavelength, no cross-lane
// traffic in the loop at all; what a lane needs of its records (direction, C of the work item's wavelengths) is staged in LDS
// by a pass in which the lanes are the tile's 64 records. A work item is (cell, group of LG wavelengths); a lane keeps
// LG x 4 accumulators (4 blocks of 16 slots = 64 slots per pass; slot tables with more slots take more passes), summed over
// the four record residues once per cell.
template <int LG>
__global__ void __launch_bounds__(64) k_log_reduce16(const DevProblem* __restrict__ Pp, const char* __restrict__ log, const unsigned long long* __restrict__ start,
                                                     const unsigned long long* __restrict__ sorted, int ncells, int ngroups, int* __restrict__ counter)
{
    const DevProblem& P = *Pp;
    __shared__ double sh_n[3][64];      // direction of the tile's records, in lane order of the compute phase
    __shared__ double sh_C[LG][64];     // C of the item's wavelengths
    __shared__ double sh_t[4][3][64];   // slot vectors of the four slot tables for the 64 slots of the pass
    const int lane = lane_id();
    const int ps = lane & 15, g = lane >> 4;
    const int L = P.L;
    for (;;) {
        int t = 0;
        if (lane == 0) t = atomicAdd(counter, 1);
        t = first_lane_i(t);
        if (t >= ncells * ngroups) break;
        const int cell = t / ngroups, lbase = (t % ngroups) * LG;
        const unsigned long long r0 = start[cell], r1 = start[cell + 1];
        if (r1 == r0) continue;
        // the item's wavelengths: constants in scalar registers
        double ga[LG], gb[LG];
        int np[LG], tid[LG];
        int npmax = 0;
        unsigned long long lmask = 0ull;
#pragma unroll
        for (int q = 0; q < LG; q++) {
            const int l = lbase + q;
            ga[q] = 1.0; gb[q] = 0.0; np[q] = 0; tid[q] = -1;
            if (l < L) {
                const double gq = P.g[l];
                tid[q] = P.tabid[l]; np[q] = P.npix[l];
                if (tid[q] >= 0) { ga[q] = 1.0 + gq * gq; gb[q] = 2.0 * gq; }   // isotropic: x = 1, weight exactly 1 (ffn = 1/npix_hp, npix_hp = 1, :2266)
                npmax = max(npmax, np[q]);
                lmask |= 1ull << l;
            }
        }
        for (int p0 = 0; p0 < npmax; p0 += 64) {
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const double* tab = P.utab[u];
                const int n = P.utab_np[u];
                const int p = p0 + lane;
                const bool ok = tab && p < n;
                sh_t[u][0][lane] = ok ? tab[p] : 0.0; sh_t[u][1][lane] = ok ? tab[n + p] : 0.0; sh_t[u][2][lane] = ok ? tab[2 * n + p] : 0.0;
            }
            double acc[LG][4];
#pragma unroll
            for (int q = 0; q < LG; q++)
#pragma unroll
                for (int b = 0; b < 4; b++) acc[q][b] = 0.0;
            // The tiles go through a pipeline so that no load is waited for where it is issued: in iteration t the record offsets
            // of tile t+2, the headers of tile t+1 and the values of tile t (C of the item's wavelengths, the direction tables'
            // rows) are requested, THEN tile t-1 is computed from LDS, and only then are tile t's values written there.
            const long long ntiles = (long long)((r1 - r0 + 63ull) >> 6);
            unsigned long long off_b = 0ull, key_b = 0ull, msk_b = 0ull, off_a = 0ull;
            if (r0 + lane < r1) off_b = sorted[r0 + lane];
            if (r0 + lane < r1) { const unsigned long long* rec = (const unsigned long long*)(log + (off_b << 3)); key_b = rec[0]; msk_b = rec[1]; }
            if (r0 + 64 + lane < r1) off_a = sorted[r0 + 64 + lane];
            unsigned long long pres[LG], present = 0ull;
            int nt_prev = 0;
#pragma unroll
            for (int q = 0; q < LG; q++) pres[q] = 0ull;
            const int pos = (lane & ~15) | ((lane & 3) << 2) | ((lane >> 2) & 3);
            for (long long tt = 0; tt <= ntiles; tt++) {
                // ---- requests (lanes = records)
                double cv[LG], st = 0.0, ct = 0.0, sp = 0.0, cp = 0.0;
                bool hasdir = false;
#pragma unroll
                for (int q = 0; q < LG; q++) cv[q] = 0.0;
                const unsigned long long base = r0 + ((unsigned long long)tt << 6);
                const int nt = (tt < ntiles) ? (int)min((unsigned long long)64, r1 - base) : 0;
                if (lane < nt) {
                    const double* rc = (const double*)(log + (off_b << 3)) + 2;
#pragma unroll
                    for (int q = 0; q < LG; q++) {
                        const int l = lbase + q;
                        if ((msk_b >> l) & 1ull) cv[q] = rc[__popcll(msk_b & ((1ull << l) - 1ull))];
                    }
                    if (msk_b & lmask) { hasdir = true; log_dir_loads(P, (int)((key_b >> 27) & 31ull), (int)(key_b >> 32), st, ct, sp, cp); }
                }
                if (tt + 1 < ntiles) {
                    off_b = off_a; key_b = 0ull; msk_b = 0ull;
                    if (base + 64 + lane < r1) { const unsigned long long* rec = (const unsigned long long*)(log + (off_b << 3)); key_b = rec[0]; msk_b = rec[1]; }
                    off_a = 0ull;
                    if (base + 128 + lane < r1) off_a = sorted[base + 128 + lane];
                }
                // ---- tile tt - 1: lanes = (slot, record residue)
                if (tt > 0 && present) {
                for (int s = 0; s < nt_prev; s += 16) {
                    if (!((present >> s) & 0xFFFFull)) continue;
                    double rn[3][4];
#pragma unroll
                    for (int k = 0; k < 3; k++)
#pragma unroll
                        for (int j = 0; j < 4; j++) rn[k][j] = sh_n[k][s + 4 * g + j];
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        if (p0 + 16 * b >= npmax) continue;
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            unsigned qu = 0u;   // the item's wavelengths that use table u (or none: u == 0 also takes the isotropic ones) and reach this block
#pragma unroll
                            for (int q = 0; q < LG; q++)
                                if (((pres[q] >> s) & 0xFFFFull) && np[q] > p0 + 16 * b && (tid[q] == u || (u == 0 && tid[q] < 0))) qu |= 1u << q;   // some record of the sub-tile holds it
                            if (!qu) continue;
                            const double tx = sh_t[u][0][16 * b + ps], ty = sh_t[u][1][16 * b + ps], tz = sh_t[u][2][16 * b + ps];
                            double cs[4];
#pragma unroll
                            for (int j = 0; j < 4; j++) cs[j] = __builtin_fma(tz, rn[2][j], __builtin_fma(ty, rn[1][j], tx * rn[0][j]));
#pragma unroll
                            for (int q = 0; q < LG; q++) {
                                if (!((qu >> q) & 1u)) continue;
                                double C[4], x[4], y[4], w[4];
#pragma unroll
                                for (int j = 0; j < 4; j++) { C[j] = sh_C[q][s + 4 * g + j]; x[j] = __builtin_fma(-gb[q], cs[j], ga[q]); }
                                // x**-1.5 = y0^3 (1 + 1.5 (1 - x y0^2)) with y0 = v_rsq_f64(x), good to e ~ 1.6e-7: the cube of the raw root
                                // corrected to first order, error 7.5 e^2 < 2e-13 (nothing decides on the weights) - six instructions and
                                // a chain of five instead of the Newton step's seven and seven (bench step 3.74 -> 3.67 s). The tracing
                                // kernel keeps the Newton form: there the shorter chain re-dealt the spills and lost (rt_wide.hpp).
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int j = 0; j < 4; j++) y[j] = __builtin_amdgcn_rsq(x[j]);
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int j = 0; j < 4; j++) w[j] = y[j] * y[j];
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int j = 0; j < 4; j++) { x[j] = __builtin_fma(-x[j], w[j], 1.0); w[j] = w[j] * y[j]; }
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int j = 0; j < 4; j++) x[j] = __builtin_fma(1.5, x[j], 1.0);
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int j = 0; j < 4; j++) w[j] = w[j] * x[j];
                                __builtin_amdgcn_sched_barrier(0);
                                acc[q][b] = __builtin_fma(C[3], w[3], __builtin_fma(C[2], w[2], __builtin_fma(C[1], w[1], __builtin_fma(C[0], w[0], acc[q][b]))));
                            }
                        }
                    }
                }
                }
                // ---- tile tt's values -> LDS (position: sub-tile, residue, j)
                __syncthreads();   // the readers of the previous tile are done (one wave per block: a wave barrier)
                sh_n[0][pos] = hasdir ? st * cp : 0.0; sh_n[1][pos] = hasdir ? st * sp : 0.0; sh_n[2][pos] = hasdir ? ct : 0.0;
                present = 0ull;
#pragma unroll
                for (int q = 0; q < LG; q++) {
                    sh_C[q][pos] = cv[q];
                    pres[q] = __ballot(cv[q] != 0.0);   // the tile's records (= lanes) that hold wavelength q of the item
                    present |= pres[q];
                }
                nt_prev = nt;
                __syncthreads();
            }
            // ---- the four record residues of a slot are added up; the lanes of residue 0 add the rows
#pragma unroll
            for (int q = 0; q < LG; q++) {
                const int l = lbase + q;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    double v = acc[q][b];
                    v += __shfl_xor(v, 16, 64);
                    v += __shfl_xor(v, 32, 64);
                    const int p = p0 + 16 * b + ps;
                    if (l < L && g == 0 && p < np[q] && v != 0.0) {
                        double* row = P.scaspe[l] + (unsigned long long)cell * (unsigned)P.nstride[l];
                        row[p] += v;
                    }
                }
            }
        }
    }
}
