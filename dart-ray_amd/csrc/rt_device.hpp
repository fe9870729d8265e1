// rt_device.hpp — device-side building blocks of the ray-cell kernels (gfx950, wave64).
//
// Everything here is written for CDNA4 directly: 64-lane ballots / readlane broadcasts, packed
// tree coordinates in 32-bit words, SoA ray queues in a per-wave HBM arena. Semantics follow the
// reference's Fortran (file:line cited per function, paths relative to /root/reference); the
// arithmetic keeps the reference's operation order and single-precision literals so that branch
// decisions match the oracle. Compile with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DRT_MAXLVL 8   // levels below the root; 4 bits per level per axis in a 32-bit word
#define DRT_MAXL 64    // wavelengths carried per ray (one lane each)

// rt_type ids (rt_routines.f90:195-217)
enum { RTT_PRECALC_CELL = 1, RTT_PRECALC_SRC = 2, RTT_DIR_CELL = 5, RTT_I_OBS_DIR_CELL = 6, RTT_DIR_SRC = 7,
       RTT_I_OBS_DIR_SRC = 8, RTT_SCATT = 10, RTT_I_OBS = 11 };
// ray types (ray_list.f90:49-59) and status ids (rt_routines.f90:255-264)
enum { RAY_HIGH = 0, RAY_LOW = 1, RAY_RECO = 2, RAY_GONE = 3, RAY_ROOT = 4 };  // ROOT: first ray of a sub-tree task
enum { ST_CONT = 0, ST_GO_HIGH = 1, ST_GONE = 2, ST_GO_LOW = 3, ST_OUT = 4 };
// error bits
enum { ERR_NEG_LENGTH = 1, ERR_CELL_NOT_FOUND = 2, ERR_CAPACITY = 4, ERR_NSIDE = 8, ERR_STEPS = 16, ERR_RELOCATE = 32, ERR_MERGE = 64, ERR_TABLE = 128,
       ERR_LOG = 256 };  // the deposit log is full (recoverable like ERR_CAPACITY: the launch is undone and repeated in more batches)

// counters[] slots
enum { CNT_INTERSECTIONS = 0, CNT_RAYS = 1, CNT_DEPOSITS = 2, CNT_SCASPE = 3, CNT_LAMDEP = 4, CNT_UADD = 5, CNT_UREAD = 6, CNT_SCASLOTS = 7,
       CNT_ATOM = 20, CNT_ATOM_LANES = 21,
       CNT_QMAX = 30,   // (free since round 3: the longest ray queue of a launch is derived from the sources' queue lengths, src_cost[N + cell])
       CNT_N = 32 };    // 8..19, 22..29: phase timers of a -DDRT_PROFILE build

struct DevProblem {
    int N, max_lvl, base0, base1, L, nsrc, ndir, pabs_max;
    int bm_par, bm_par_max, wall_on[3];
    double wall[6];
    double modelsize, rad_lim_ms, en_lim, cs;
    double csize_arr[DRT_MAXLVL + 1], carea_arr[DRT_MAXLVL + 1], cvol_arr[DRT_MAXLVL + 1];
    // constants with the reference's precision (see oracle/dartray_oracle.c)
    double halfpi, pi, twopi, twothird, glepsilon, tau_thin, len_eps, outcube_f, merge_tol;
    // tree (row G of SURVEY.md section 8a)
    const int* cchild;
    const int* lvl;
    const double* ccoord;    // [N][3]
    const double* csize;     // [N]
    const uint32_t* cpack;   // [N][3]  tree coordinates, 4 bits per level (from cindex)
    // Face neighbours of every leaf: nbr[6*cc + 2*axis + (direction > 0)] = the cell the reference's step
    // (rt_routines.f90:1778-1841: carry the +-1 up the levels, then ccindd_to_cc from the root) ends in when it does not
    // have to descend below the level of cc: bits 0-26 cell, bits 27-30 its level, bit 31 set when that cell is
    // subdivided (the ray then descends with the coordinates of the finer levels); 0xFFFFFFFF: the ray leaves the model.
    // One load instead of a walk of lvl(cc) dependent loads from the root.
    const uint32_t* nbr;
    // fields, wavelength fastest: [N][L]
    const double* dens;
    double* u_fest;
    double* u_final;
    const double* lumcell;
    const double* p_src_pos;  // [nsrc][3]
    const double* p_src_lum;  // [nsrc][L]
    const int* src_cell;      // [N]
    const double* dir_out;    // [2][ndir]
    double* lum_lost;         // [L]
    double* i_obs;            // [ndir][N+nsrc][L]
    // per wavelength
    double ksca[DRT_MAXL], g[DRT_MAXL];
    int npix[DRT_MAXL], npix_hp[DRT_MAXL], kp[DRT_MAXL];
    int nstride[DRT_MAXL];                // row length of the scaspe arrays: npix rounded up to 8 doubles, so that a row starts on
                                          // a 64-byte line and an atomic instruction over it is 7 memory-side requests (55 slots), not 8
    double* scaspe[DRT_MAXL];             // [N][nstride]
    const double* scaspe_prev[DRT_MAXL];
    const double* scaspe_tot[DRT_MAXL];
    const double* tab[DRT_MAXL];          // [3][npix]: unit vectors of the scaspe slots (x, y, z); null if isotropic
    int tabid[DRT_MAXL];                  // index of the slot table (kp_unique order), -1 if isotropic
    const double* utab[4];                // the unique slot tables, utab_np[] slots each
    int utab_np[4];
    // per-wave ray-queue arena
    char* arena;
    unsigned long long* counters;
    int* error;
    int* task_counter;
    double* src_cost;          // [2N]: shader-clock cycles the waves spent on each source's tasks in this launch (steers the next one);
                               // [N + cell]: the longest ray queue among the source's tasks, as the kernel leaves it; the host scales it to
                               // 48 sectors per task after the launch (rt_loop: scale_queues)
    // Direction tables built on the HOST with the host libm (see build_dir_tables in dartray_rt.hip):
    // per HEALPix level k (nside = 2^k) ring table [4*nside][3] = (theta, sin, cos) indexed by ring jr,
    // and azimuth tables [..][3] = (phi, sin, cos): kshift=0 rows at 2*nr*(nr-1) + jp-1, kshift=1 rows
    // (equatorial rings only) at jp-1. Pixel centres then carry bit-identical sines and cosines on
    // host and device, which keeps every geometric branch of the path identical to the CPU result.
    const double* ring_tab[14];
    const double* phi0_tab[14];
    const double* phi1_tab[14];
    int tab_max_level;         // largest k with tables; finer levels fall back to device sin/cos/acos
    // HG normalisation of calc_ffn_arr (rt_routines.f90:2252-2260), sum over the HEALPix slots of x**-1.5, tabulated per
    // ray direction: hgnorm[k] = [12*4^k pixels][L] for nside = 2^k <= 2^hgnorm_max (1.0 for isotropic wavelengths). A
    // ray direction is a pixel centre, the sum depends on nothing else: one load per (ray, wavelength) deposit
    // replaces 48 (or 12) x**-1.5 per wavelength of every launched ray. Finer rays compute it at launch as before.
    const double* hgnorm[14];
    int hgnorm_max;
    const double* dir_out_sc;  // [ndir][4] sin(theta), cos(theta), sin(phi), cos(phi) of the observer directions
    // optional debug trace of ray launches: dbg[0] = count, then records of 6 ints
    int* dbg;
    int dbg_cap;
    int iobs_healpix;  // rt_algorithm 'i_obs' (rt_routines.f90:1538-1548): the scattered light towards an observer is read from the HEALPix
                       // pixel of scaspe_tot that holds the direction (assign_src_lum), not from the run's own observer slot
    int dbg_flags;  // developer builds (-DDRT_DEBUG_FLAGS): bit 0 = skip the scaspe / u_final atomics of the deposit loop (timing only)
};

// ------------------------------------------------------------------------------------------------
// wave helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ unsigned long long lanemask_lt() { return (1ull << lane_id()) - 1ull; }
__device__ __forceinline__ int bcast_i(int v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ double bcast_d(double v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ int first_lane_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
// Orders the global-memory traffic of the lanes of ONE wave: stores of some lanes (queue appends, parked rays, hash
// slots) are read by other lanes after it. A fence at WORKGROUP scope: s_waitcnt vmcnt(0) and nothing else. That is what
// the AMDGPU memory model prescribes for a release/acquire pair inside one work-group (gfx90a/gfx942 family, work-groups
// not split across CUs): the waves of a work-group share their CU's write-through vector L1, a store updates it on its
// way to L2, so a later load of the same CU cannot see an older copy - and the lanes of one wave are the narrowest case
// of that. The agent-scope fence used before (buffer_wbl2 + buffer_inv: write back and invalidate the XCD's L2 and the
// L1) costs microseconds each, tens of them per task, and the L2 write-backs of 3 000 waves queue behind one another:
// with it a task carried ~0.3 ms of fixed cost (TRUST-I tau = 0.01: 72 ms at one sector per task against 15 ms at 16).
#ifndef DRT_FENCE_SCOPE
#define DRT_FENCE_SCOPE "workgroup"
#endif
__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, DRT_FENCE_SCOPE); }
// OR over the wave. The result steers wave-uniform control flow, so it is handed back through readfirstlane: a value
// the compiler can prove uniform lives in a scalar register (a uniform value kept in a vector register can be spilled
// inside divergent code, where only the active lanes are saved).
__device__ __forceinline__ int wave_or(int v)
{
    for (int off = 1; off < 64; off <<= 1) v |= __shfl_xor(v, off, 64);
    return __builtin_amdgcn_readfirstlane(v);
}

// ------------------------------------------------------------------------------------------------
// HEALPix NESTED, 32-bit (healpix_routines.f90:30-136, 238-348). The reference's 1024-/128-entry
// lookup tables (mk_pix2xy :145, mk_xy2pix1 :351) are bit (de)interleaves; done here with shifts.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t compact_even_bits(uint32_t v)
{
    v &= 0x55555555u;
    v = (v | (v >> 1)) & 0x33333333u;
    v = (v | (v >> 2)) & 0x0F0F0F0Fu;
    v = (v | (v >> 4)) & 0x00FF00FFu;
    v = (v | (v >> 8)) & 0x0000FFFFu;
    return v;
}
__device__ __forceinline__ uint32_t spread_bits(uint32_t v)
{
    v &= 0x0000FFFFu;
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

// Pixel centre with its sines and cosines. Uses the host-built tables when the level has them.
struct PixDir { double theta, phi, st, ct, sp, cp; };
// Returns false when `nside` is finer than the host-built tables (the caller raises ERR_TABLE).
__device__ inline bool pix2ang_nest_sc(const DevProblem& P, int nside, int ipix, PixDir& o)
{
    const int k = 31 - __clz(nside);
    if (k > P.tab_max_level) { o.theta = o.phi = 0; o.st = o.sp = 0; o.ct = o.cp = 1; return false; }
    const int jrll[12] = {2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4};
    const int jpll[12] = {1, 3, 5, 7, 0, 2, 4, 6, 1, 3, 5, 7};
    int npface = nside * nside, nl4 = 4 * nside;
    int face_num = ipix / npface;
    int ipf = ipix - face_num * npface;
    int ix = (int)compact_even_bits((uint32_t)ipf);
    int iy = (int)compact_even_bits((uint32_t)ipf >> 1);
    int jrt = ix + iy, jpt = ix - iy;
    int jr = jrll[face_num] * nside - jrt - 1;
    int nr, kshift;
    if (jr < nside) { nr = jr; kshift = 0; }
    else if (jr <= 3 * nside) { nr = nside; kshift = (jr - nside) & 1; }
    else { nr = nl4 - jr; kshift = 0; }
    int jp = (jpll[face_num] * nr + jpt + 1 + kshift) / 2;
    if (jp > nl4) jp -= nl4;
    if (jp < 1) jp += nl4;
    const double* rt = P.ring_tab[k] + 3 * jr;
    o.theta = rt[0]; o.st = rt[1]; o.ct = rt[2];
    const double* pt = (kshift == 0) ? (P.phi0_tab[k] + 3ull * (2ull * nr * (nr - 1) + (jp - 1))) : (P.phi1_tab[k] + 3ull * (jp - 1));
    o.phi = pt[0]; o.sp = pt[1]; o.cp = pt[2];
    return true;
}

// Exact remainder of non-negative doubles, a - trunc(a/b)*b (Fortran MOD, rt_routines.f90:1836), without
// the library's fmod: one fma gives the exact remainder for the rounded quotient, then at most one
// correction step (both exact, the remainder of two doubles is representable).
// x**(-1.5) for the Henyey-Greenstein weights (calc_ffn_arr, rt_routines.f90:2254): hardware
// reciprocal square root (1.6e-7) refined by one Newton step, then cubed: 1.3e-14 relative
// (tools/micro/rsq_test.hip); the same function normalises the weights, so most of that cancels.
__device__ __forceinline__ double pow_m1p5(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - (0.5 * x) * y * y);
    return y * y * y;
}

__device__ __forceinline__ double fmod_pos(double a, double b)
{
    double q = trunc(a / b);
    double r = __builtin_fma(-q, b, a);
    if (r < 0) r += b;
    else if (r >= b) r -= b;
    return r;
}

// ang2pix_nest with z = cos(theta) supplied by the caller (healpix_routines.f90:269)
__device__ inline int ang2pix_nest_z(const DevProblem& P, int nside, double z, double phi);

__device__ inline int ang2pix_nest_z(const DevProblem& P, int nside, double z, double phi)
{
    double za = fabs(z);
    double m = fmod(phi, P.twopi);
    if (m < 0) m += P.twopi;
    double tt = m / P.halfpi;
    int face_num, ix, iy;
    if (za <= P.twothird) {
        int jp = (int)(nside * (0.5 + tt - z * 0.75));
        int jm = (int)(nside * (0.5 + tt + z * 0.75));
        int ifp = jp / nside, ifm = jm / nside;
        if (ifp == ifm) face_num = (ifp & 3) + 4;
        else if (ifp < ifm) face_num = (ifp & 3);
        else face_num = (ifm & 3) + 8;
        ix = jm & (nside - 1);
        iy = nside - (jp & (nside - 1)) - 1;
    } else {
        int ntt = (int)tt;
        if (ntt >= 4) ntt = 3;
        double tp = tt - ntt;
        double tmp = sqrt(3.0 * (1.0 - za));
        int jp = (int)(nside * tp * tmp);
        int jm = (int)(nside * (1.0 - tp) * tmp);
        if (jp > nside - 1) jp = nside - 1;
        if (jm > nside - 1) jm = nside - 1;
        if (z >= 0) { face_num = ntt; ix = nside - jm - 1; iy = nside - jp - 1; }
        else { face_num = ntt + 8; ix = jp; iy = jm; }
    }
    int ipf = (int)(spread_bits((uint32_t)ix) | (spread_bits((uint32_t)iy) << 1));
    return ipf + face_num * nside * nside;
}

// ------------------------------------------------------------------------------------------------
// tree coordinates: ccindd(3,max_lvl) packed as one 32-bit word per axis, level j in bits 4(j-1)..
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int nib_get(uint32_t w, int j) { return (int)((w >> (4 * (j - 1))) & 15u); }
__device__ __forceinline__ uint32_t nib_set(uint32_t w, int j, int v)
{
    int s = 4 * (j - 1);
    return (w & ~(15u << s)) | (((uint32_t)v & 15u) << s);
}

// ccindd_to_cc (smooth_grid_routines.f90:820-852) with a bounds guard: a lookup that would leave
// the cell arrays reports "not found" (the reference aborts there, rt_routines.f90:1851).
__device__ inline bool ccindd_to_cc(const DevProblem& P, uint32_t cx, uint32_t cy, uint32_t cz, int& cc, int& clvl)
{
    int l = 1;
    for (int i = 1; i <= P.max_lvl; i++) {
        int b = (i > 1) ? P.base1 : P.base0;
        int k = (nib_get(cz, i) * b + nib_get(cy, i)) * b + nib_get(cx, i);
        l += k;
        if (l < 1 || l >= P.N) return false;
        int ch = P.cchild[l];
        if (ch == -1) { cc = l; clvl = i; return true; }
        l = ch;
    }
    return false;
}
