// rt_kernels.hpp — the ray-cell kernels (gfx950).
//
// Mapping: one persistent WAVEFRONT owns one task = (source, group of HEALPix
// sectors) and runs that task's whole nside ladder (main_dir_loop, rt_routines.f90:1343-1467)
// without leaving the wave. Inside the wave a lane is one (ray, wavelength) pair: LP = next power
// of two >= lnum lanes share a ray, so a wave carries 64/LP rays. Geometry is evaluated by every
// lane of a ray (same instructions, no extra issue slots); intensities, optical depths and
// deposits are per lane. Queues ("high" = to split, "low" = to merge, ray_list.f90) live as SoA
// arrays in a per-wave HBM arena and are appended with ballot + prefix-popcount compaction.
// The P-wide scattered-light axpy (process_scatt_rad, rt_routines.f90:2638) is wave-cooperative:
// the depositing lane's cell / energy / direction are broadcast with readlane and the 64 lanes
// become the P scaspe slots of that cell -> one contiguous f64 atomic instruction per 64 slots.
#pragma once
#include "rt_device.hpp"

struct StageArgs {
    int rt_type;
    int nsources;          // number of emitting cells in src_list
    const int* src_list;   // cell ids
    int sectors_per_task;  // G: 48 / G tasks per source
    int scatter;           // 1 for dir/scatt stages (scaspe deposits), 0 for precalc
    // per-launch carve-up of the ray-queue arena: the fewer waves a stage uses, the deeper its queues
    int cap, hcap;
    unsigned long long wave_stride;
    // Sub-tree tasks (few, deep sources): phase 1 (export_level = 4) traces the 192 nside-4 rays of a
    // source and exports the blocked ones; phase 2 (spawn_count > 0) runs one task per child pixel of
    // an exported ray. Rays at nside <= 8 are never queued for merging (rt_routines.f90:2494), so the
    // ladders below distinct nside-8 pixels never exchange rays: 768 independent tasks per source
    // instead of 48 sectors, each performing the reference's operations in the reference's order.
    int export_level;
    int spawn_count;
    char* spawn;        // records: SoA list of spawn_cap rays + int src[spawn_cap]
    int spawn_cap;
    int* spawn_counter;
    // Several ranks share the (source, sector) tasks of a point-source stage: this rank runs the tasks
    // task_first + k * task_stride (rt_routines.f90:1356-1358: the 48 sectors of a source are independent)
    int task_first, task_stride;
    // Cell sources that cost the waves most in the previous launch (DevProblem::src_cost) are cut into smaller tasks, which
    // come first: heavy[t] = (cell, first sector | sectors << 8) for t < n_heavy; the tasks of the other sources follow,
    // sectors_per_task sectors each. A task is one wave's serial work: no task may be a large part of the whole launch.
    const int2* heavy;
    int n_heavy;
    // print_psel_av (rt_routines.f90:1456-1465): sum and number of the paths of the rays of every source that ended in this
    // stage ([N + nsrc] each, null = off); the average is taken on the host (after the ranks' sums were added, if any)
    double* psel_sum;
    double* psel_cnt;
    int record_cost;    // add every task's duration to DevProblem::src_cost[cell] (cell-source stages)
    // phase-alignment window of the ray-major kernel in cell sizes (0 = every busy ray steps every time), see trace_items_w
    double phase_win;
    // deposit log (rt_stage_wide<LP, 2>, see rt_wide.hpp): log = log_nchunks chunks of log_chunk_bytes; log_idx = per chunk
    // log_chunk_idx index words (cell | byte offset / 8 << 27, one per record); log_chunk_count[k] = records of chunk k;
    // *log_chunks_used = chunks handed out so far
    char* log;
    unsigned long long* log_idx;
    unsigned* log_chunk_count;
    unsigned* log_cell_count;   // [N]: records per cell of the launch
    int* log_chunks_used;
    unsigned log_chunk_bytes, log_chunk_idx;
    int log_nchunks;
};

struct ListView {
    uint32_t* ipix;  // HEALPix pixel (NESTED) of the ray direction at the ray's own nside
    uint32_t* meta;  // log2(nside) | isel << 8 | ray_type << 12 | merged << 16
    int* cc;         // cc_old: cell where the ray was blocked (source cell for a first launch)
    double* prev;    // path already covered
    double* dpl;     // dplane
    double* lum;     // [cap][LP] equivalent source luminosity per wavelength
};

__device__ __forceinline__ ListView list_view(char* abase, unsigned long long lbytes, int cap, int LP, int list)
{
    char* b = abase + (unsigned long long)list * lbytes;
    ListView v;
    v.ipix = (uint32_t*)b;
    v.meta = (uint32_t*)(b + 4ull * cap);
    v.cc = (int*)(b + 8ull * cap);
    v.prev = (double*)(b + 12ull * cap + 4ull * (cap & 1));
    v.dpl = v.prev + cap;
    v.lum = v.dpl + cap;
    return v;
}
__host__ __device__ inline unsigned long long list_bytes(int cap, int LP)
{
    return 12ull * cap + 4ull * (cap & 1) + 16ull * cap + 8ull * (unsigned long long)cap * LP;
}

// Layout of ONE wave's slice of the ray-queue arena (byte offsets from the slice's base). One definition for the kernels
// that carve the slice and for the host that sizes it (carve_arena) - and for the test that checks that no region
// overlaps the next and that the slice ends inside its stride (tests/test_abi.py).
//   list[0..3]  SoA ray queues of `cap` rays: high A, high B, low, staging
//   hkey        [hcap] int            sibling hash: parent pixel
//   hmem        [4*hcap + cap] int    sibling slots of every hash entry, then one code per extracted ray
//   ostat       [cap] int             status of parked rays / prefix weights
//   thg         [LP][64] double       HG normalisation of every (wavelength, ray) of the ray-major kernel
//   inten       [LP][64] double       ray intensities when they are not kept in LDS (-DDRT_I_GLOBAL)
struct ArenaLayout { unsigned long long list[4], hkey, hmem, ostat, thg, inten, end; };
__host__ __device__ inline ArenaLayout arena_layout(int cap, int hcap, int LP)
{
    ArenaLayout a;
    const unsigned long long lb = list_bytes(cap, LP);
    for (int k = 0; k < 4; k++) a.list[k] = (unsigned long long)k * lb;
    a.hkey = 4ull * lb;
    a.hmem = a.hkey + 4ull * (unsigned long long)hcap;
    a.ostat = a.hmem + 4ull * (4ull * (unsigned long long)hcap + (unsigned long long)cap);
    a.thg = (a.ostat + 4ull * (unsigned long long)cap + 15ull) & ~15ull;
    a.inten = a.thg + 512ull * (unsigned long long)LP;
    a.end = a.inten + 512ull * (unsigned long long)LP;
    return a;
}

#define META(nsl, isel, type, merged) ((uint32_t)(nsl) | ((uint32_t)(isel) << 8) | ((uint32_t)(type) << 12) | ((uint32_t)(merged) << 16))
#define META_NSL(m) ((int)((m) & 31u))
#define META_ISEL(m) ((int)(((m) >> 8) & 3u))
#define META_TYPE(m) ((int)(((m) >> 12) & 7u))
#define META_MERGED(m) ((int)(((m) >> 16) & 1u))

struct SrcInfo {  // wave-uniform description of the emitting cell / point source of the current task
    int nc, lvl_nc;
    double sx, sy, sz;       // src_ccoord
    double csize_i, area_i;  // csize_arr(lvl(nc)), carea_arr(lvl(nc))
    double norm;             // 4*pi*csize(nc)**2 (rt_routines.f90:1673)
    uint32_t px, py, pz;     // ccindd_nc
};

struct Ray {
    int cc, clvl, clvl_old, cc_new, isel_old, ipix_dir;
    bool flag_beam, lam_active;
    uint32_t cx, cy, cz;
    int incs;  // the direction's sign along x, y, z (-1, 0, 1), two bits each (+1): one register instead of three took the
               // one-wavelength step of rt_stage_wide from 632 to 587 ms (bit-fields for clvl / clvl_old / isel_old / flags: no gain)
    __device__ __forceinline__ int inc(int k) const { return (int)(((unsigned)incs >> (2 * k)) & 3u) - 1; }
    __device__ __forceinline__ void set_inc(int k, int v) { incs = (int)(((unsigned)incs & ~(3u << (2 * k))) | ((unsigned)(v + 1) << (2 * k))); }
    double cpi0, cpi1, cpi2;  // 1/cproj (rt_routines.f90:1671)
    double prev;
    double I;                 // ray_intensity of this lane's wavelength
    double tot_hg;            // HG normalisation: sum over the HEALPix slots of x**-1.5 (calc_ffn_arr, rt_routines.f90:2258)
    double nx, ny, nz;        // unit vector of the direction used by calc_ads_arr (:2294)
};

// value of lane `src` (wave-uniform) of a double, through SGPRs
__device__ __forceinline__ double readlane_d(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int src)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(v & 0xffffffffull), src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(v >> 32), src);
    return ((unsigned long long)hi << 32) | lo;
}

// global-address loads and atomics: FLAT instructions (generic pointers) count in lgkmcnt as well as vmcnt, so with
// flat atomics in flight every LDS or scalar wait also waits for the atomics (see rt_wide.hpp)
typedef __attribute__((address_space(1))) double gdouble;
typedef __attribute__((address_space(1))) const double gcdouble;
__device__ __forceinline__ double gload(const double* p) { return *(gcdouble*)p; }
__device__ __forceinline__ void gatomic_add(double* p, double v)
{
#ifdef DRT_NO_ATOMICS
    asm volatile("" ::"v"(p), "v"(v));
#else
    (void)__hip_atomic_fetch_add((gdouble*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

template <class T> struct WaveCountersT { T inter, rays, deps, sca, lamdep, uadd, uread, slots;
    unsigned psel_n;                      // wave-uniform: rays of the running task that ended (psel_av)
    unsigned long long atom, atom_lanes;  // wave-uniform: atomic wave-instructions of the deposit loop and their active lanes
#ifdef DRT_PROFILE
    unsigned long long t_setup, t_step, t_flush, t_park, t_next, t_commit, t_extract, t_total, n_steps, n_busy_lanes, n_groups, n_deps,
        t_dep, t_ghead, t_memb, t_epi, n_lam, n_memb, n_cls0, n_cls1, n_cls2;  // inside the wavelength loop: deposit arithmetic, group head, member loop, epilogue + atomic
#endif
};
typedef WaveCountersT<unsigned long long> WaveCounters;
// rt_stage_wide: the eight per-lane tallies in 32 bits (8 registers fewer in a kernel that spills), moved to the 64-bit
// counters in memory at the end of a task once a lane has passed 2^30
#ifndef DRT_WIDE_CNT_T
#define DRT_WIDE_CNT_T unsigned
#endif
typedef WaveCountersT<DRT_WIDE_CNT_T> WideCounters;
#ifdef DRT_PROFILE
#define TSTAMP(v) unsigned long long v = __builtin_amdgcn_s_memtime()
#define TACC(field, t0, t1) wc.field += (t1) - (t0)
#else
#define TSTAMP(v)
#define TACC(field, t0, t1)
#endif

// calc_psel (rt_routines.f90:2307-2351)
__device__ __forceinline__ void calc_psel(const DevProblem& P, const SrcInfo& S, Ray& r, double& psel, int& isel, double& c0, double& c1, double& c2)
{
    double pmax = (double)P.pabs_max;
    double p0 = pmax, p1 = pmax, p2 = pmax;
    const double* c = P.ccoord + 3ull * (unsigned long long)r.cc;
    c0 = c[0]; c1 = c[1]; c2 = c[2];
    // The reference caches pabs_arr = cpi * (-src + 0.5 * inc * csize) while the ray stays at one level and then uses
    // cpi * c + pabs_arr, which rounds differently from the direct form. The cached value is a function of (cpi, src, inc,
    // csize(level)) alone, so it is recomputed here bit for bit instead of being kept in six registers per ray.
    const double cs = P.csize_arr[r.clvl];
    if (r.clvl == r.clvl_old) {
        if (r.inc(0) != 0) p0 = r.cpi0 * c0 + r.cpi0 * ((-S.sx) + 0.5 * (double)r.inc(0) * cs);
        if (r.inc(1) != 0) p1 = r.cpi1 * c1 + r.cpi1 * ((-S.sy) + 0.5 * (double)r.inc(1) * cs);
        if (r.inc(2) != 0) p2 = r.cpi2 * c2 + r.cpi2 * ((-S.sz) + 0.5 * (double)r.inc(2) * cs);
    } else {
        if (r.inc(0) != 0) p0 = r.cpi0 * (c0 - S.sx + 0.5 * (double)r.inc(0) * cs);
        if (r.inc(1) != 0) p1 = r.cpi1 * (c1 - S.sy + 0.5 * (double)r.inc(1) * cs);
        if (r.inc(2) != 0) p2 = r.cpi2 * (c2 - S.sz + 0.5 * (double)r.inc(2) * cs);
    }
    if (p0 < p2) {
        if (p0 < p1) { psel = p0; isel = 1; } else { psel = p1; isel = 2; }
    } else {
        if (p2 < p1) { psel = p2; isel = 3; } else { psel = p1; isel = 2; }
    }
}

__device__ __forceinline__ void calc_psel(const DevProblem& P, const SrcInfo& S, Ray& r, double& psel, int& isel)
{
    double c0, c1, c2;
    calc_psel(P, S, r, psel, isel, c0, c1, c2);
}

// find_cc_new2 + calc_outcube (rt_routines.f90:1962-2085). d = ray direction (not inverted).
// returns 0 = located, 1 = outside the model, 2 = error
__device__ inline int find_cc_new2(const DevProblem& P, const SrcInfo& S, Ray& r, int cc_old, double d0, double d1,
                                   double d2, double prev_ray)
{
    double pc0 = S.sx + prev_ray * d0, pc1 = S.sy + prev_ray * d1, pc2 = S.sz + prev_ray * d2;
    double half = P.modelsize / 2.;
    if (fabs(pc0) > half || fabs(pc1) > half || fabs(pc2) > half) return 1;
    const double* c = P.ccoord + 3ull * (unsigned long long)cc_old;
    int lv_old = P.lvl[cc_old];
    double cellsize = P.csize[cc_old];
    double lim = P.outcube_f * cellsize / 2.0;
    if (!(fabs(pc0 - c[0]) > lim) && !(fabs(pc1 - c[1]) > lim) && !(fabs(pc2 - c[2]) > lim)) {
        r.clvl = lv_old;
        r.cc_new = cc_old;
        // cindex_to_ccindd fills levels 1..clvl; deeper levels are rewritten before they are read
        const uint32_t* cp = P.cpack + 3ull * (unsigned long long)cc_old;
        r.cx = cp[0]; r.cy = cp[1]; r.cz = cp[2];
        return 0;
    }
    double n0 = pc0 + half, n1 = pc1 + half, n2 = pc2 + half;
    uint32_t cx = 0, cy = 0, cz = 0;
    for (int i = 1; i <= P.max_lvl; i++) {
        int b = (i > 1) ? P.base1 : P.base0;
        double cs = P.csize_arr[i];
        int v0 = (int)(n0 / cs), v1 = (int)(n1 / cs), v2 = (int)(n2 / cs);
        if (v0 == b) v0 = b - 1;
        if (v1 == b) v1 = b - 1;
        if (v2 == b) v2 = b - 1;
        cx = nib_set(cx, i, v0); cy = nib_set(cy, i, v1); cz = nib_set(cz, i, v2);
        n0 = n0 - v0 * cs; n1 = n1 - v1 * cs; n2 = n2 - v2 * cs;
    }
    int ccn, lv;
    if (!ccindd_to_cc(P, cx, cy, cz, ccn, lv)) return 2;
    const double* cn = P.ccoord + 3ull * (unsigned long long)ccn;
    lim = P.outcube_f * P.csize_arr[lv] / 2.0;
    if ((fabs(pc0 - cn[0]) > lim) || (fabs(pc1 - cn[1]) > lim) || (fabs(pc2 - cn[2]) > lim)) return 2;
    r.cx = cx; r.cy = cy; r.cz = cz;
    r.cc_new = ccn;
    r.clvl = lv;
    return 0;
}

// Step the packed tree coordinates across face `isel` and re-derive the finer levels of the other
// two axes (rt_routines.f90:1778-1841, increment smooth_grid_routines.f90:751-769).
// Returns false when the ray leaves the model.
__device__ __forceinline__ bool step_axis(const DevProblem& P, uint32_t& w, int inc, bool is_sel, int clvl, double psel,
                                          double cpi, double s, double c)
{
    if (is_sel) {
        int k = inc;
        for (int j = clvl; j >= 1; j--) {
            int b = (j > 1) ? P.base1 : P.base0;
            int v = nib_get(w, j) + k;
            int ovr = k;
            if (v >= b) v = 0;
            else if (v < 0) v = b - 1;
            else ovr = 0;
            w = nib_set(w, j, v);
            k = ovr;
            if (k == 0) break;
        }
        if (k != 0) return false;
        for (int j = clvl + 1; j <= P.max_lvl; j++) {
            int b = (j > 1) ? P.base1 : P.base0;
            if (inc == 1) w = nib_set(w, j, 0);
            else if (inc == -1) w = nib_set(w, j, b - 1);
        }
    } else {
        double cellsize = P.csize_arr[clvl];
        double bb = psel / cpi + (s - c + 0.5 * cellsize);
        for (int j = clvl + 1; j <= P.max_lvl; j++) {
            int b = (j > 1) ? P.base1 : P.base0;
            cellsize = P.csize_arr[j];
            int v = (int)(bb / cellsize);
            if (v == b) v = b - 1;
            w = nib_set(w, j, v);
            bb = (bb >= 0) ? fmod_pos(bb, cellsize) : -fmod_pos(-bb, cellsize);
        }
    }
    return true;
}

// The step into the next cell (rt_routines.f90:1778-1851) through the face-neighbour table. Same cell as step_axis +
// ccindd_to_cc give: the neighbour at the level of the old cell (or the coarser leaf that covers it) comes from the
// table; where finer cells lie behind the face the tree coordinates of the levels below are derived exactly as the
// reference does (:1807-1837: 0 / base-1 on the crossed axis, int(b/csize_arr(j)) with the clamp and the MOD on the others).
// cold: coordinates of the old cell's centre. Returns false when the walk fails (:1851); inside = false: ray left the model.
#define DRT_NBR_OUT 0xFFFFFFFFu
__device__ __forceinline__ bool next_cell(const DevProblem& P, const SrcInfo& S, Ray& r, int isel, double psel, double c0, double c1, double c2,
                                          bool& inside)
{
    const int inc = (isel == 1) ? r.inc(0) : (isel == 2) ? r.inc(1) : r.inc(2);
    const uint32_t e = P.nbr[6ull * (unsigned long long)r.cc + 2 * (isel - 1) + (inc > 0 ? 1 : 0)];
    inside = (e != DRT_NBR_OUT);
    if (!inside) return true;
    int l = (int)(e & 0x07FFFFFFu);
    const int lv = (int)((e >> 27) & 15u);
    if (!(e >> 31)) { r.cc = l; r.clvl = lv; return true; }
    // finer cells behind the face (the table entry is at the old cell's own level)
    double cellsize = P.csize_arr[r.clvl];
    double b0 = psel / r.cpi0 + (S.sx - c0 + 0.5 * cellsize);
    double b1 = psel / r.cpi1 + (S.sy - c1 + 0.5 * cellsize);
    double b2 = psel / r.cpi2 + (S.sz - c2 + 0.5 * cellsize);
    for (int j = lv + 1; j <= P.max_lvl; j++) {
        const int ch = P.cchild[l];
        if (ch == -1) { r.cc = l; r.clvl = j - 1; return true; }
        const int b = (j > 1) ? P.base1 : P.base0;
        cellsize = P.csize_arr[j];
        int v0, v1, v2;
        if (isel == 1) v0 = (inc == 1) ? 0 : b - 1;
        else { v0 = (int)(b0 / cellsize); if (v0 == b) v0 = b - 1; b0 = (b0 >= 0) ? fmod_pos(b0, cellsize) : -fmod_pos(-b0, cellsize); }
        if (isel == 2) v1 = (inc == 1) ? 0 : b - 1;
        else { v1 = (int)(b1 / cellsize); if (v1 == b) v1 = b - 1; b1 = (b1 >= 0) ? fmod_pos(b1, cellsize) : -fmod_pos(-b1, cellsize); }
        if (isel == 3) v2 = (inc == 1) ? 0 : b - 1;
        else { v2 = (int)(b2 / cellsize); if (v2 == b) v2 = b - 1; b2 = (b2 >= 0) ? fmod_pos(b2, cellsize) : -fmod_pos(-b2, cellsize); }
        l = ch + (v2 * b + v1) * b + v0;
        if (l < 1 || l >= P.N) return false;
    }
    if (P.cchild[l] != -1) return false;
    r.cc = l; r.clvl = P.max_lvl;
    return true;
}

// check_wall_hit (rt_routines.f90:2572-2596)
__device__ __forceinline__ bool wall_hit(const DevProblem& P, int rt, int cc)
{
    if (rt == RTT_DIR_SRC) return false;
    const double* c = P.ccoord + 3ull * (unsigned long long)cc;
    bool hit = false;
    if (P.wall_on[0] && (c[0] < P.wall[0] || c[0] > P.wall[1])) hit = true;
    if (P.wall_on[1] && (c[1] < P.wall[2] || c[1] > P.wall[3])) hit = true;
    if (P.wall_on[2] && (c[2] < P.wall[4] || c[2] > P.wall[5])) hit = true;
    return hit;
}

// deposit (rt_routines.f90:2362-2557) for one (ray, wavelength) lane. `grp` is the lane mask of the
// ray's wavelengths; the ballot for any(ray_intensity > 0) is taken over it. Returns the status,
// the scattered energy en_sca of this lane (to be spread over scaspe by the wave) and updates
// r.I / lum_lost. `live` = lane holds a real wavelength of a busy ray.
__device__ __forceinline__ int deposit(const DevProblem& P, int rt, const SrcInfo& S, Ray& r, bool live, int il, int level,
                                       double length, double psel, unsigned long long grp, double& en_sca,
                                       double& u_add, double& lum_lost_acc, WaveCounters& wc)
{
    en_sca = 0.0;
    u_add = 0.0;
    const int cc = r.cc, nc = S.nc;
    const bool act = live && r.lam_active;
    const unsigned long long idx = (unsigned long long)cc * P.L + il;
    double dens = act ? P.dens[idx] : 0.0;
    if (act) wc.lamdep++;
    double tau_c = dens * length;
    double ex = exp(-tau_c);
    int npix_beam = 12 * (level * level);
    double beam_i = 4 * P.pi / npix_beam;
    int lv = P.lvl[cc];
    double csize_f = P.csize_arr[lv], vol_f = P.cvol_arr[lv];
    double beam_f = P.carea_arr[lv] / (psel * psel);
    double I = r.I, i_av;
    if (cc != nc) {
        if (tau_c < P.tau_thin || I == 0) i_av = I;
        else i_av = I / tau_c * (1 - ex);
    } else {
        if (tau_c < P.tau_thin || I == 0) i_av = I;
        else if (rt != RTT_PRECALC_SRC && rt != RTT_DIR_SRC) {
            tau_c = dens * S.csize_i;
            ex = exp(-tau_c);
            i_av = I / (tau_c * tau_c) * (ex + tau_c - 1);
        } else {
            i_av = I / (dens * length) * (1 - ex);
        }
    }
    if (rt == RTT_PRECALC_SRC || rt == RTT_PRECALC_CELL) {
        if (psel > P.rad_lim_ms) return ST_GONE;
        if ((beam_f / P.bm_par <= beam_i) && !r.flag_beam) return ST_GO_HIGH;
        if ((beam_f / P.bm_par_max < beam_i) || (level <= 8) || r.flag_beam) {
            if (act) {
                double v = i_av * length / P.cs / vol_f * S.area_i * beam_i;
                if (v != 0.0) { atomicAdd(&P.u_fest[idx], v); wc.uadd++; }
            }
        } else return ST_GO_LOW;
    } else {
        if ((beam_f / P.bm_par <= beam_i) && !r.flag_beam) {
            if (act) {
                double u_beam = i_av * csize_f / P.cs / vol_f * S.area_i * beam_f;
                wc.uread++;
                if (u_beam < P.en_lim * P.u_fest[idx]) {
                    I = 0;
                    lum_lost_acc = lum_lost_acc + i_av * beam_i * S.csize_i * S.csize_i;
                }
                r.I = I;
            }
            unsigned long long alive = __ballot(act && I > 0) & grp;
            bool wh = wall_hit(P, rt, cc);
            if (alive != 0ull && !wh) return ST_GO_HIGH;
            return ST_GONE;
        } else if ((beam_f / P.bm_par_max < beam_i) || (level <= 8) || r.flag_beam) {
            if (act) {
                double v = i_av * length / P.cs / vol_f * S.area_i * beam_i;
                u_add = v;
                if (v != 0.0) wc.uadd++;
                double k = P.ksca[il];
                if (cc != nc) {
                    if (tau_c > P.tau_thin) en_sca = I * (1.0 - ex) * k * S.area_i * beam_i;
                    else en_sca = I * tau_c * k * S.area_i * beam_i;
                } else {
                    if (tau_c > P.tau_thin) en_sca = I / (tau_c) * (ex + tau_c - 1) * k * S.area_i * beam_i;
                    else en_sca = I * tau_c / 2 * k * S.area_i * beam_i;
                }
            }
        } else return ST_GO_LOW;
    }
    // update ray_intensity (rt_routines.f90:2543-2555)
    if (act) {
        if (cc != nc) I = I * ex;
        else if (!(tau_c < P.tau_thin)) I = I / (tau_c) * (1 - ex);
        r.I = I;
    }
    return ST_CONT;
}

// Wave-cooperative process_scatt_rad (rt_routines.f90:2638-2726) + u_final update (:2497-2505).
// For every lane holding a deposit, its cell / wavelength / energy / direction are broadcast with
// readlane and the 64 lanes become the scaspe slots of that cell: lanes 0..npix-1 add
// en_sca * weight(slot) to scaspe(:,cc) (contiguous f64 atomics, one instruction), lane 63 adds the
// energy-density increment to u_final(il,cc) in the same instruction. Slot unit vectors of the
// first two slot tables live in registers (rtab*), so a round touches no memory but the atomics.
// Weights: w_p = x_p**-1.5 / sum_q x_q**-1.5, x = 1 + g^2 - 2 g cos(angle) (calc_ffn_arr :2252-2260).
struct RegTabs {
    double x0, y0, z0, x1, y1, z1;  // slot unit vectors of the first two slot tables, slot = lane
    // per-wavelength constants of THIS lane's wavelength (lane % LP); a deposit round fetches them from
    // the depositing lane with readlane instead of loading DevProblem arrays (no memory in the round)
    int np, ns, tid;   // slots, row stride, slot table
    double ga, gb, ksca;
    double* srow;
    // read once per kernel: after an atomic the compiler must assume *Pp changed and would load these again
    double* u_final;
    int L;
};
#ifdef DRT_NO_ATOMICS  // diagnostic build: keep the values alive, skip the memory operation
#define FLUSH_ATOMIC(addr, v) asm volatile("" ::"v"(addr), "v"(v))
#else
#define FLUSH_ATOMIC(addr, v) atomicAdd(addr, v)
#endif

template <int LP>
__device__ __forceinline__ void flush_deposits(const DevProblem& P, const Ray& r, const RegTabs& T, double e, double u_add,
                                               WaveCounters& wc)
{
    const int lane = lane_id();
    unsigned long long pend = __ballot(e != 0.0 || u_add != 0.0);
#ifdef DRT_PROFILE
    {   // statistics: how many distinct (cell, wavelength) targets do this step's deposits have?
        unsigned long long q = pend;
        while (q) {
            const int s0 = __ffsll((long long)q) - 1;
            const int c0 = __builtin_amdgcn_readlane(r.cc, s0);
            q &= ~__ballot(r.cc == c0 && (lane % LP) == (s0 % LP));
            wc.n_groups++;
        }
        wc.n_deps += __popcll(pend);
    }
#endif
    const bool mine = (e != 0.0 || u_add != 0.0);
    const int il = lane % LP;
    while (pend) {
        const int src = __ffsll((long long)pend) - 1;
        const int il_b = src % LP;
        const int cc_b = __builtin_amdgcn_readlane(r.cc, src);
        // every lane of this step that deposits into the same (cell, wavelength) is summed on chip and
        // leaves the wave as one read-modify-write
        unsigned long long grp = __ballot(mine && r.cc == cc_b && il == il_b);
        pend &= ~grp;
        const int np = __builtin_amdgcn_readlane(T.np, src);
        const unsigned long long sbase = readlane_u64((unsigned long long)T.srow, src);
        double* srow = (double*)sbase + (unsigned long long)cc_b * __builtin_amdgcn_readlane(T.ns, src);
        double* uaddr = T.u_final + (unsigned long long)cc_b * T.L + il_b;
        const int tid = __builtin_amdgcn_readlane(T.tid, src);
        const double ga = readlane_d(T.ga, src), gb = readlane_d(T.gb, src);
        if (np <= 63 && tid <= 1) {
            const double px = tid == 1 ? T.x1 : T.x0, py = tid == 1 ? T.y1 : T.y0, pz = tid == 1 ? T.z1 : T.z0;
            double acc = 0.0, su = 0.0;
            while (grp) {
                const int m = __ffsll((long long)grp) - 1;
                grp &= grp - 1;
                const double e_b = readlane_d(e, m);
                su += readlane_d(u_add, m);
                if (e_b != 0.0) {
                    wc.sca++;
                    wc.slots += (unsigned long long)np;
                    if (tid < 0) acc += e_b;  // isotropic: ffn = 1/npix_hp with npix_hp = 1 (:2266)
                    else {
                        const double nx = readlane_d(r.nx, m), ny = readlane_d(r.ny, m), nz = readlane_d(r.nz, m);
                        const double C = e_b / readlane_d(r.tot_hg, m);
                        acc += C * pow_m1p5(ga - gb * (px * nx + py * ny + pz * nz));
                    }
                }
            }
            double* addr = (lane == 63) ? uaddr : &srow[lane];
            const double v = (lane == 63) ? su : acc;
            const bool go = (lane < np || lane == 63) && v != 0.0;
            const unsigned long long gb_ = __ballot(go);
            if (gb_) { wc.atom++; wc.atom_lanes += (unsigned long long)__popcll(gb_); }
            if (go) gatomic_add(addr, v);
        } else {
            const double* tab = P.tab[il_b];
            double su = 0.0;
            for (int p0 = 0; p0 < np; p0 += 64) {
                const int p = p0 + lane;
                double px = 0, py = 0, pz = 0;
                if (tab && p < np) { px = tab[p]; py = tab[np + p]; pz = tab[2 * np + p]; }
                double acc = 0.0;
                unsigned long long gm = grp;
                while (gm) {
                    const int m = __ffsll((long long)gm) - 1;
                    gm &= gm - 1;
                    const double e_b = readlane_d(e, m);
                    if (p0 == 0) {
                        su += readlane_d(u_add, m);
                        if (e_b != 0.0) { wc.sca++; wc.slots += (unsigned long long)np; }
                    }
                    if (e_b != 0.0) {
                        if (!tab) acc += e_b;
                        else {
                            const double nx = readlane_d(r.nx, m), ny = readlane_d(r.ny, m), nz = readlane_d(r.nz, m);
                            const double C = e_b / readlane_d(r.tot_hg, m);
                            acc += C * pow_m1p5(ga - gb * (px * nx + py * ny + pz * nz));
                        }
                    }
                }
                const bool go = p < np && acc != 0.0;
                const unsigned long long gb_ = __ballot(go);
                if (gb_) { wc.atom++; wc.atom_lanes += (unsigned long long)__popcll(gb_); }
                if (go) gatomic_add(&srow[p], acc);
            }
            if (__ballot(lane == 63 && su != 0.0)) { wc.atom++; wc.atom_lanes++; }
            if (lane == 63 && su != 0.0) gatomic_add(uaddr, su);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Trace all rays derived from the entries of one queue at HEALPix resolution `level`
// (the body of main_dir_loop's ext_list loop + ray_tracing, rt_routines.f90:1378-1433, 1618-1857).
// mode 0: entries of `in` are "high"/"reco" rays: 4 children (or the ray itself) are launched.
// mode 1: entries flagged `merged` are launched as they are (merged "low" rays).
//
// Queue ORDER is part of the reference's semantics: the first of four mergeable siblings in queue
// order lends its cc_old / isel / dplane to the merged ray (ray_list.f90:221-269). The reference
// traces rays one after the other, so its queues are ordered by work item (position of the parent
// in the extracted list, then child pixel). Rays finish out of order here; a blocked ray is
// therefore parked in a staging queue at the index of its item and the staging queue is compacted
// into the high / low queues in item order once the chunk of items is done.
// ------------------------------------------------------------------------------------------------
#ifndef DRT_REFILL_QUARTERS
#define DRT_REFILL_QUARTERS 4  // idle ray slots are refilled when at least this many quarters of the wave wait. Whole
                               // batches win: rays of a batch stay in phase and leave the wave in fewer atomics (TRUST-I, 1
                               // wavelength, 21 801 cells: 1: 256 ms, 2: 242, 3: 220, 4: 212)
#endif
template <int LP>
__device__ void trace_items(const DevProblem& P, const StageArgs& a, const SrcInfo& S, int mode, const ListView in,
                            int n_in, const ListView outH, int& nH, const ListView outL, int& nL, const ListView stg,
                            int* __restrict__ ostat, int level, const RegTabs& T, double& lum_lost_acc, WaveCounters& wc, int& err)
{
    constexpr int R = 64 / LP;
    constexpr unsigned long long GRP0 = (LP == 64) ? ~0ull : ((1ull << LP) - 1ull);
    const int lane = lane_id();
    const int slot = lane / LP, il = lane % LP;
    const bool lam_ok = il < P.L;
    const bool leader = (il == 0);
    const unsigned long long grp = GRP0 << (slot * LP);
    const unsigned long long below = (1ull << (slot * LP)) - 1ull;  // lanes of lower slots
    unsigned long long leaders = 0;
    for (int s = 0; s < R; s++) leaders |= 1ull << (s * LP);
    const int rt = a.rt_type;
    const int n_items = (mode == 0) ? 4 * n_in : n_in;
    const int nsl = 31 - __clz(level);
    bool busy = false;
    int my_item = 0;
    Ray r;
    r.cc = 0; r.clvl = 1; r.clvl_old = -1; r.cc_new = -1; r.isel_old = 0; r.ipix_dir = 0; r.flag_beam = true;
    r.lam_active = false; r.cx = r.cy = r.cz = 0; r.incs = 0x15; r.cpi0 = r.cpi1 = r.cpi2 = 1.0;
    r.prev = 0; r.I = 0; r.tot_hg = 1.0; r.nx = r.ny = r.nz = 0;
    long long guard = 0;

    for (int chunk_base = 0; chunk_base < n_items; chunk_base += a.cap) {
        const int chunk_end = min(n_items, chunk_base + a.cap);
        for (int i = lane; i < chunk_end - chunk_base; i += 64) ostat[i] = 0;
        wave_fence();
        int next_item = chunk_base;

        for (;;) {
            const unsigned long long busy_b = __ballot(busy);
            const int n_busy = __popcll(busy_b & leaders);
            const bool more = next_item < chunk_end;
            if (!more && n_busy == 0) break;
            if (++guard > (1ll << 26)) { err |= ERR_STEPS; break; }

            TSTAMP(ts0);
            // ---- refill idle ray slots -----------------------------------------------------------
            if (more && (n_busy == 0 || 4 * (R - n_busy) >= DRT_REFILL_QUARTERS * R)) {
                const unsigned long long idle_leaders = (~busy_b) & leaders;
                const int my_rank = __popcll(idle_leaders & below);
                const int item = next_item + my_rank;
                next_item += __popcll(idle_leaders);
                if (!busy && item < chunk_end) {
                    const int parent = (mode == 0) ? (item >> 2) : item;
                    const int child = (mode == 0) ? (item & 3) : 0;
                    const uint32_t m = in.meta[parent];
                    const int type = META_TYPE(m);
                    const bool valid = (mode == 0) ? (type == RAY_HIGH || ((type == RAY_RECO || type == RAY_ROOT) && child == 0)) : (META_MERGED(m) != 0);
                    if (valid) {
                        const bool is_reco = (type == RAY_RECO);
                        int ipix_dir = (int)in.ipix[parent];
                        if (mode == 0 && type == RAY_HIGH) ipix_dir = 4 * ipix_dir + child;
                        PixDir pd;
                        if (!pix2ang_nest_sc(P, level, ipix_dir, pd)) err |= ERR_TABLE;
                        const double theta = pd.theta, phi = pd.phi;
                        const int cc_old = in.cc[parent];
                        const double prev_ray = in.prev[parent];
                        double src_lum = lam_ok ? in.lum[(unsigned long long)parent * LP + il] : 0.0;
                        if (rt == RTT_SCATT && cc_old == S.nc && lam_ok) {
                            // assign_src_lum (rt_routines.f90:3043-3071)
                            int kp = P.kp[il];
                            int ip = (kp >= 0) ? ang2pix_nest_z(P, 1 << kp, pd.ct, phi) : 0;
                            src_lum = P.scaspe_prev[il][(unsigned long long)S.nc * P.nstride[il] + ip] * P.npix_hp[il];
                        }
                        const double sin_dl = pd.st, cos_dl = pd.ct, sin_al = pd.sp, cos_al = pd.cp;
                        const double d0 = sin_dl * cos_al, d1 = sin_dl * sin_al, d2 = cos_dl;  // calc_cproj :1945
                        (void)theta;
                        bool ok = true;
                        if (leader) {
                            wc.rays++;  // a call of ray_tracing, including those that return at once (:1656)
                            if (P.dbg) {
                                int k = atomicAdd(&P.dbg[0], 1);
                                if (k < P.dbg_cap) {
                                    int* d = P.dbg + 1 + 6 * k;
                                    d[0] = S.nc; d[1] = level; d[2] = ipix_dir; d[3] = cc_old; d[4] = (type == RAY_ROOT) ? RAY_HIGH : type; d[5] = item;
                                }
                            }
                        }
                        if (cc_old == S.nc) {
                            r.cc = S.nc; r.prev = 0.0; r.isel_old = 0; r.cx = S.px; r.cy = S.py; r.cz = S.pz;
                            r.cc_new = -1; r.clvl = S.lvl_nc;
                        } else {
                            int fr = find_cc_new2(P, S, r, cc_old, d0, d1, d2, prev_ray);
                            if (fr == 2) err |= ERR_RELOCATE;
                            if (fr != 0) ok = false;
                            r.prev = prev_ray; r.cc = r.cc_new; r.isel_old = META_ISEL(m);
                        }
                        if (ok) {
                            r.flag_beam = true;
                            r.clvl_old = -1;
                            r.ipix_dir = ipix_dir;
                            // incvec (rt_routines.f90:1666-1669)
                            int v;
                            v = (int)(d0 / P.glepsilon); r.set_inc(0, max(min(v, 1), -1));
                            v = (int)(d1 / P.glepsilon); r.set_inc(1, max(min(v, 1), -1));
                            v = (int)(d2 / P.glepsilon); r.set_inc(2, max(min(v, 1), -1));
                            r.cpi0 = 1 / d0; r.cpi1 = 1 / d1; r.cpi2 = 1 / d2;
                            r.I = src_lum / S.norm;
                            r.lam_active = lam_ok && (r.I > 0);
                            r.tot_hg = 1.0;
                            if (a.scatter) {
                                // calc_ffn_arr (rt_routines.f90:2186-2281); "reco" rays read table row 1
                                double nx = d0, ny = d1, nz = d2;
                                if (is_reco && level <= 256) {
                                    PixDir p1;
                                    pix2ang_nest_sc(P, level, 1, p1);
                                    nx = p1.st * p1.cp; ny = p1.st * p1.sp; nz = p1.ct;
                                }
                                r.nx = nx; r.ny = ny; r.nz = nz;
                                const double* tab = lam_ok ? P.tab[il] : nullptr;
                                if (r.lam_active && tab && nsl <= P.hgnorm_max) {
                                    // tabulated per direction (DevProblem::hgnorm); "reco" rays use pixel 1 like their direction
                                    r.tot_hg = P.hgnorm[nsl][(unsigned long long)((is_reco && level <= 256) ? 1 : ipix_dir) * P.L + il];
                                } else if (r.lam_active && tab) {
                                    const double g = P.g[il];
                                    const int np = P.npix[il], hp = P.npix_hp[il];
                                    const double ga = 1.0 + g * g, gb = 2.0 * g;
                                    double tot = 0;
                                    for (int p = 0; p < hp; p++) {
                                        double ads = tab[p] * nx + tab[np + p] * ny + tab[2 * np + p] * nz;
                                        tot += pow_m1p5(ga - gb * ads);
                                    }
                                    r.tot_hg = tot;
                                }
                            }
                            busy = true;
                            my_item = item - chunk_base;
                        }
                    }
                }
            }

            TSTAMP(ts1);
            TACC(t_setup, ts0, ts1);
#ifdef DRT_PROFILE
            wc.n_steps++; wc.n_busy_lanes += __popcll(__ballot(busy));
#endif
            // ---- one ray-cell intersection for every busy slot ---------------------------------------
            int status = ST_CONT;
            double en_sca = 0.0, u_add = 0.0, psel = 0.0;
            int isel = 0;
            if (busy) {
                if (leader) wc.inter++;
                calc_psel(P, S, r, psel, isel);
                double length = psel - r.prev;
                if (fabs(length) < P.len_eps * P.csize_arr[r.clvl]) length = 0;
                if (length < 0) { err |= ERR_NEG_LENGTH; status = ST_GONE; }
                if (r.cc != S.nc && r.cc != r.cc_new) r.flag_beam = false;
                if (length != 0. && status == ST_CONT) {
                    if (leader) wc.deps++;
                    status = deposit(P, rt, S, r, lam_ok, il, level, length, psel, grp, en_sca, u_add, lum_lost_acc, wc);
                }
            }
            TSTAMP(ts2);
            TACC(t_step, ts1, ts2);
            if (a.scatter) flush_deposits<LP>(P, r, T, en_sca, u_add, wc);
            TSTAMP(ts3);
            TACC(t_flush, ts2, ts3);

            // ---- blocked rays are parked at their item's index (add_to_high/low_ray_list, ray_list.f90:463-579)
            if (busy && (status == ST_GO_HIGH || status == ST_GO_LOW)) {
                const bool hi = (status == ST_GO_HIGH);
                if (leader) {
                    const double dpl = r.prev / ((r.isel_old == 1) ? r.cpi0 : (r.isel_old == 2) ? r.cpi1 : r.cpi2);
                    stg.ipix[my_item] = (uint32_t)r.ipix_dir;
                    stg.meta[my_item] = META(nsl, r.isel_old, hi ? RAY_HIGH : RAY_LOW, 0);
                    stg.cc[my_item] = r.cc;
                    // add_to_high zeroes prev/dplane when isel_old == 0 (ray_list.f90:496-503)
                    stg.prev[my_item] = (hi && r.isel_old == 0) ? 0.0 : r.prev;
                    stg.dpl[my_item] = (hi && r.isel_old == 0) ? 0.0 : dpl;
                    ostat[my_item] = hi ? 1 : 2;
                }
                if (lam_ok) stg.lum[(unsigned long long)my_item * LP + il] = r.I * S.norm;
            }
            if (busy && status != ST_CONT) busy = false;
            TSTAMP(ts4);
            TACC(t_park, ts3, ts4);

            // ---- next cell --------------------------------------------------------------------------------
            if (busy) {
                r.prev = psel;
                r.isel_old = isel;
                r.clvl_old = r.clvl;
                const double* c = P.ccoord + 3ull * (unsigned long long)r.cc;
                bool inside = true;
                if (!next_cell(P, S, r, isel, psel, c[0], c[1], c[2], inside)) { err |= ERR_CELL_NOT_FOUND; busy = false; }
                if (!inside) busy = false;
            }
            TSTAMP(ts5);
            TACC(t_next, ts4, ts5);
        }
        wave_fence();
        TSTAMP(tc0);

        // ---- commit the parked rays in item order ------------------------------------------------------
        for (int base = 0; base < chunk_end - chunk_base; base += 64) {
            const int i = base + lane;
            const int st = (i < chunk_end - chunk_base) ? ostat[i] : 0;
            const unsigned long long bh = __ballot(st == 1), bl = __ballot(st == 2);
            if (!(bh | bl)) continue;
            const int pos = (st == 1) ? nH + __popcll(bh & lanemask_lt()) : nL + __popcll(bl & lanemask_lt());
            if (st != 0) {
                if (pos < a.cap) {
                    const ListView& o = (st == 1) ? outH : outL;
                    o.ipix[pos] = stg.ipix[i]; o.meta[pos] = stg.meta[i]; o.cc[pos] = stg.cc[i];
                    o.prev[pos] = stg.prev[i]; o.dpl[pos] = stg.dpl[i];
                    for (int k = 0; k < P.L; k++) o.lum[(unsigned long long)pos * LP + k] = stg.lum[(unsigned long long)i * LP + k];
                } else err |= ERR_CAPACITY;
            }
            nH += __popcll(bh);
            nL += __popcll(bl);
        }
        wave_fence();
        TSTAMP(tc1);
        TACC(t_commit, tc0, tc1);
        if (err & ERR_STEPS) break;
    }
}

// ------------------------------------------------------------------------------------------------
// extract_ray_list, "low" branch (ray_list.f90:167-298): pull the rays with nside == 2*level out of
// the low queue (stable), group the siblings of each coarser pixel through a hash table keyed by the
// parent pixel (NESTED: parent = ipix >> 2, slot = ipix & 3; rays tile the sphere, so a slot is held
// by at most one ray), merge complete groups that crossed the same plane, and re-queue the rest as
// "reco" rays in the order the reference's i/j loops would (:221-296). One lane per queue entry.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool sib_match(double d_i, int isel_i, double d_j, int isel_j, double tol)
{
    return (isel_i == isel_j) && (fabs(d_i - d_j) / fabs(d_i) < tol);
}

template <int LP>
__device__ void extract_low(const DevProblem& P, const StageArgs& a, const ListView low, int& nL, const ListView ext, int& nE,
                            const ListView high, int& nH, int* hkey, int* hmem, int* __restrict__ wts, int level, int& err)
{
    const int lane = lane_id();
    const int want = (31 - __clz(level)) + 1;  // log2(2*level)
    const int L = P.L;
    int w = 0;
    nE = 0;
    for (int base = 0; base < nL; base += 64) {
        const int i = base + lane;
        const bool valid = i < nL;
        uint32_t m = 0, ip = 0; int cc = 0; double pv = 0, dp = 0;
        if (valid) { m = low.meta[i]; ip = low.ipix[i]; cc = low.cc[i]; pv = low.prev[i]; dp = low.dpl[i]; }
        const bool isx = valid && META_NSL(m) == want;
        const bool keep = valid && !isx;
        const unsigned long long bx = __ballot(isx), bk = __ballot(keep);
        const int posx = nE + __popcll(bx & lanemask_lt());
        const int posk = w + __popcll(bk & lanemask_lt());
        for (int k = 0; k < L; k++) {
            double v = valid ? low.lum[(unsigned long long)i * LP + k] : 0.0;
            if (isx) ext.lum[(unsigned long long)posx * LP + k] = v;
            if (keep && posk != i) low.lum[(unsigned long long)posk * LP + k] = v;
        }
        if (isx) { ext.meta[posx] = m; ext.ipix[posx] = ip; ext.cc[posx] = cc; ext.prev[posx] = pv; ext.dpl[posx] = dp; }
        if (keep && posk != i) { low.meta[posk] = m; low.ipix[posk] = ip; low.cc[posk] = cc; low.prev[posk] = pv; low.dpl[posk] = dp; }
        nE += __popcll(bx);
        w += __popcll(bk);
    }
    nL = w;
    if (nE == 0) return;
    // hash table sized to the next power of two >= 2*nE
    int hsize = 64;
    while (hsize < 2 * nE) hsize <<= 1;
    if (hsize > a.hcap) { err |= ERR_CAPACITY; nE = 0; return; }
    const int hmask = hsize - 1;
    for (int h = lane; h < hsize; h += 64) { hkey[h] = -1; hmem[4 * h] = -1; hmem[4 * h + 1] = -1; hmem[4 * h + 2] = -1; hmem[4 * h + 3] = -1; }
    wave_fence();
    for (int base = 0; base < nE; base += 64) {
        const int i = base + lane;
        if (i < nE) {
            const int ip = (int)ext.ipix[i];
            const int key = ip >> 2;
            int h = (int)(((uint32_t)key * 2654435761u) >> 7) & hmask;
            for (int probe = 0; probe < hsize; probe++) {
                int old = atomicCAS(&hkey[h], -1, key);
                if (old == -1 || old == key) break;
                h = (h + 1) & hmask;
            }
            if (atomicCAS(&hmem[4 * h + (ip & 3)], -1, i) != -1) err |= ERR_MERGE;  // two rays on one pixel: tiling broken
        }
    }
    wave_fence();
    const int nsl = 31 - __clz(level);
    // pass 1: classify every extracted ray. wts[i] = number of "reco" rays that the reference appends
    // when its i loop reaches entry i (0 if i is merged, already collected, or not a group leader).
    // The per-entry result (leader index << 3 | rank << 1 | merged-leader) is kept in hkey-independent
    // storage: the high bits of ext.meta are free, but a second int array keeps the code plain.
    for (int base = 0; base < nE; base += 64) {
        const int i = base + lane;
        int wt = 0;
        if (i < nE) {
            const int ip = (int)ext.ipix[i];
            const int key = ip >> 2;
            int h = (int)(((uint32_t)key * 2654435761u) >> 7) & hmask;
            for (int probe = 0; probe < hsize; probe++) {
                int kk = __atomic_load_n(&hkey[h], __ATOMIC_RELAXED);
                if (kk == key || kk == -1) break;
                h = (h + 1) & hmask;
            }
            int mm[4];
            int n = 0;
            for (int q = 0; q < 4; q++) { int v = __atomic_load_n(&hmem[4 * h + q], __ATOMIC_RELAXED); if (v >= 0) mm[n++] = v; }
            // sort the (<= 4) members by queue position
            for (int x = 1; x < n; x++) for (int y = x; y > 0 && mm[y - 1] > mm[y]; y--) { int t = mm[y]; mm[y] = mm[y - 1]; mm[y - 1] = t; }
            double dd[4]; int is[4];
            for (int q = 0; q < n; q++) { dd[q] = ext.dpl[mm[q]]; is[q] = META_ISEL(ext.meta[mm[q]]); }
            // replay the reference's loops on this sibling set
            bool gone[4] = {false, false, false, false};
            int my_lead = -1, my_rank = 0;
            bool my_merged = false;
            for (int x = 0; x < n; x++) {
                if (gone[x]) continue;
                int cnt = 1, memb[4];
                memb[0] = x;
                for (int y = x + 1; y < n && cnt < 4; y++)
                    if (!gone[y] && sib_match(dd[x], is[x], dd[y], is[y], P.merge_tol)) memb[cnt++] = y;
                const bool mrg = (cnt == 4);
                for (int q = 0; q < cnt; q++) {
                    gone[memb[q]] = true;
                    if (mm[memb[q]] == i) { my_lead = mm[x]; my_rank = q; my_merged = mrg; }
                }
                if (mm[x] == i && !mrg) wt = cnt;
            }
            // stash: merged flag + leader + rank in hmem-independent scratch (wts holds the weight; the
            // rest goes to the staging fields of ext that the merge does not need any more: none) ->
            // keep it in registers via a second array
            if (my_lead < 0) { err |= ERR_MERGE; my_lead = i; }
            hmem[4 * hsize + i] = (my_lead << 3) | (my_rank << 1) | (my_merged ? 1 : 0);
        }
        if (i < nE) wts[i] = wt;
    }
    wave_fence();
    // pass 2: exclusive prefix sum of wts -> first position of each leader's group in the high queue
    int run = nH;
    for (int base = 0; base < nE; base += 64) {
        const int i = base + lane;
        int v = (i < nE) ? wts[i] : 0;
        int incl = v;
        for (int off = 1; off < 64; off <<= 1) { int t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
        if (i < nE) wts[i] = run + incl - v;
        run += __builtin_amdgcn_readlane(incl, 63);  // scalar: `run` becomes the new queue length
    }
    const int nH_new = run;
    wave_fence();
    // pass 3: write reco rays to their slots; merged leaders rewrite their own entry
    for (int base = 0; base < nE; base += 64) {
        const int i = base + lane;
        if (i < nE) {
            const int code = hmem[4 * hsize + i];
            const int lead = code >> 3, rank = (code >> 1) & 3;
            const bool mrg = code & 1;
            const uint32_t m = ext.meta[i];
            const int ip = (int)ext.ipix[i];
            if (!mrg) {
                const int pos = wts[lead] + rank;
                if (pos < a.cap) {
                    high.ipix[pos] = (uint32_t)ip;
                    high.meta[pos] = META(nsl, META_ISEL(m), RAY_RECO, 0);
                    high.cc[pos] = ext.cc[i];
                    high.prev[pos] = ext.prev[i];
                    high.dpl[pos] = ext.dpl[i];
                    for (int k = 0; k < L; k++) high.lum[(unsigned long long)pos * LP + k] = ext.lum[(unsigned long long)i * LP + k];
                } else err |= ERR_CAPACITY;
            }
        }
    }
    wave_fence();
    for (int base = 0; base < nE; base += 64) {
        const int i = base + lane;
        if (i < nE) {
            const int code = hmem[4 * hsize + i];
            const int lead = code >> 3;
            const bool mrg = code & 1;
            const uint32_t m = ext.meta[i];
            const int ip = (int)ext.ipix[i];
            if (mrg && lead == i) {
                // merged ray: parent direction, mean luminosity and path (ray_list.f90:255-269); the
                // other three members are the remaining slots of this pixel's hash entry
                const int key = ip >> 2;
                int h = (int)(((uint32_t)key * 2654435761u) >> 7) & hmask;
                for (int probe = 0; probe < hsize; probe++) {
                    int kk = hkey[h];
                    if (kk == key || kk == -1) break;
                    h = (h + 1) & hmask;
                }
                int mm[4] = {hmem[4 * h], hmem[4 * h + 1], hmem[4 * h + 2], hmem[4 * h + 3]};
                for (int x = 1; x < 4; x++) for (int y = x; y > 0 && mm[y - 1] > mm[y]; y--) { int t = mm[y]; mm[y] = mm[y - 1]; mm[y - 1] = t; }
                double pv = (ext.prev[mm[0]] + ext.prev[mm[1]] + ext.prev[mm[2]] + ext.prev[mm[3]]) / 4;
                for (int k = 0; k < L; k++) {
                    double v = ext.lum[(unsigned long long)mm[0] * LP + k];
                    v = v + ext.lum[(unsigned long long)mm[1] * LP + k];
                    v = v + ext.lum[(unsigned long long)mm[2] * LP + k];
                    v = v + ext.lum[(unsigned long long)mm[3] * LP + k];
                    ext.lum[(unsigned long long)i * LP + k] = v / 4;
                }
                ext.prev[i] = pv;
                ext.ipix[i] = (uint32_t)(ip >> 2);
                ext.meta[i] = META(nsl, META_ISEL(m), RAY_LOW, 1);
            }
        }
    }
    wave_fence();
    // the other members of merged groups and the re-queued rays are "gone" (ray_list.f90:269, 289)
    for (int base = 0; base < nE; base += 64) {
        const int i = base + lane;
        if (i < nE) {
            const int code = hmem[4 * hsize + i];
            if (!((code & 1) && (code >> 3) == i)) {
                const uint32_t m = ext.meta[i];
                ext.meta[i] = META(META_NSL(m), META_ISEL(m), RAY_GONE, 0);
            }
        }
    }
    nH = nH_new;
    if (nH > a.cap) { err |= ERR_CAPACITY; nH = a.cap; }
    wave_fence();
}

// ------------------------------------------------------------------------------------------------
// rt_loop (rt_routines.f90:985-1058) + main_dir_loop (:1343-1467): persistent waves pull tasks.
// ------------------------------------------------------------------------------------------------
// Waves per SIMD the register allocator is told to make room for. The kernel wants ~370 registers;
// half the time of a lone wave is SQ_WAIT_ANY (dependent tree loads, loads queued behind the f64
// atomics in vmcnt order), so trading registers for resident waves pays until the spills reach the
// hot loop. Measured on TRUST-I tau=1 (profiles/r01_occupancy.txt): 64 rays per wave (LP = 1) is best
// at 2 waves/SIMD (1: 262 ms, 2: 237, 3: 267, 4: 292); 2 rays per wave (LP = 32) at 4
// (1: 5225 ms, 2: 3140, 3: 2732, 4: 2504, 5: 3557).
#ifdef DRT_MIN_WAVES
#define DRT_WAVES_FOR(LP) DRT_MIN_WAVES
#else
#define DRT_WAVES_FOR(LP) ((LP) >= 8 ? 4 : (LP) >= 2 ? 3 : 2)
#endif
template <int LP>
__device__ __forceinline__ void rt_stage_body(const DevProblem* __restrict__ Pp, const StageArgs& a)
{
    const DevProblem& P = *Pp;
    const int lane = lane_id();
    const int il = lane % LP;
    const int waves_per_block = blockDim.x >> 6;
    // wave-uniform by construction, and told so: the arena pointers derived from it live in scalar registers
    const int gwave = blockIdx.x * waves_per_block + first_lane_i((int)(threadIdx.x >> 6));
    char* abase = P.arena + (unsigned long long)gwave * a.wave_stride;
    const unsigned long long lbytes = list_bytes(a.cap, LP);
    const ArenaLayout AL = arena_layout(a.cap, a.hcap, LP);
    int* hkey = (int*)(abase + AL.hkey);
    int* hmem = (int*)(abase + AL.hmem);
    int* ostat = (int*)(abase + AL.ostat);
    const int G = a.sectors_per_task;
    const int tasks_per_src = 48 / G;
    const bool from_spawn = a.spawn_count > 0;
    const int ntasks = from_spawn ? 4 * a.spawn_count : a.n_heavy + a.nsources * tasks_per_src;
    const ListView SP = list_view(a.spawn, 0, a.spawn_cap, LP, 0);
    const int* spawn_src = (const int*)(a.spawn + list_bytes(a.spawn_cap, LP));
    const int rt = a.rt_type;
    WaveCounters wc = {};
    int err = 0;
    TSTAMP(tk0);
    // slot unit vectors of the first two scaspe slot tables, one slot per lane
    RegTabs T = {0, 0, 0, 0, 0, 0, 0, 0, -1, 1.0, 0.0, 0.0, nullptr, P.u_final, P.L};
    if (a.scatter && il < P.L) {
        const double g = P.g[il];
        T.np = P.npix[il]; T.ns = P.nstride[il]; T.tid = P.tabid[il]; T.ga = 1.0 + g * g; T.gb = 2.0 * g; T.ksca = P.ksca[il]; T.srow = P.scaspe[il];
    }
    if (a.scatter) {
        if (P.utab[0] && lane < P.utab_np[0]) { const int n = P.utab_np[0]; T.x0 = P.utab[0][lane]; T.y0 = P.utab[0][n + lane]; T.z0 = P.utab[0][2 * n + lane]; }
        if (P.utab[1] && lane < P.utab_np[1]) { const int n = P.utab_np[1]; T.x1 = P.utab[1][lane]; T.y1 = P.utab[1][n + lane]; T.z1 = P.utab[1][2 * n + lane]; }
    }

    for (;;) {
        int task = 0;
        if (lane == 0) task = __atomic_load_n(P.error, __ATOMIC_RELAXED) ? 0x7fffffff : a.task_first + a.task_stride * atomicAdd(P.task_counter, 1);  // stop at the first error anywhere
        task = first_lane_i(task);
        if (task >= ntasks) break;
        const unsigned long long t_task = __builtin_amdgcn_s_memtime();
        int nc, sec0 = 0, Gt = G;
        if (from_spawn) nc = spawn_src[task >> 2];
        else if (task < a.n_heavy) { const int2 h = a.heavy[task]; nc = h.x; sec0 = h.y & 255; Gt = h.y >> 8; }
        else { const int t = task - a.n_heavy; nc = a.src_list[t / tasks_per_src]; sec0 = (t % tasks_per_src) * G; }
        SrcInfo S;
        S.nc = nc;
        S.lvl_nc = P.lvl[nc];
        if (rt == RTT_PRECALC_SRC || rt == RTT_DIR_SRC) {  // calc_src_ccoord (rt_routines.f90:1922-1939)
            const double* ps = P.p_src_pos + 3ull * (unsigned long long)P.src_cell[nc];
            S.sx = ps[0]; S.sy = ps[1]; S.sz = ps[2];
        } else {
            const double* c = P.ccoord + 3ull * (unsigned long long)nc;
            S.sx = c[0]; S.sy = c[1]; S.sz = c[2];
        }
        S.csize_i = P.csize_arr[S.lvl_nc];
        S.area_i = P.carea_arr[S.lvl_nc];
        S.norm = 4 * P.pi * P.csize[nc] * P.csize[nc];
        S.px = P.cpack[3ull * nc]; S.py = P.cpack[3ull * nc + 1]; S.pz = P.cpack[3ull * nc + 2];

        // create_high_ray_list (ray_list.f90:71-106) for the G sectors of this task
        ListView Hc = list_view(abase, lbytes, a.cap, LP, 0);
        ListView Hn = list_view(abase, lbytes, a.cap, LP, 1);
        const ListView LW = list_view(abase, lbytes, a.cap, LP, 2);
        const ListView ST = list_view(abase, lbytes, a.cap, LP, 3);
        double src_lum = 0.0;
        if (il < P.L) {
            if (rt == RTT_PRECALC_SRC || rt == RTT_DIR_SRC) src_lum = P.p_src_lum[(unsigned long long)P.src_cell[nc] * P.L + il];
            else src_lum = P.lumcell[(unsigned long long)nc * P.L + il];
        }
        int nH = Gt, nL = 0, level = 4;
        if (!from_spawn) {
            for (int s = lane / LP; s < Gt; s += 64 / LP) {
                if (il == 0) {
                    Hc.ipix[s] = (uint32_t)(sec0 + s);  // idir*4 + idir2 at nside 2
                    Hc.meta[s] = META(1, 0, RAY_HIGH, 0);
                    Hc.cc[s] = nc;
                    Hc.prev[s] = 0.0;
                    Hc.dpl[s] = 0.0;
                }
                Hc.lum[(unsigned long long)s * LP + il] = src_lum;
            }
        } else {
            // sub-tree task: child (task & 3) of the exported nside-4 ray, launched at nside 8
            const int rec = task >> 2;
            if (lane < LP) {
                if (il == 0) {
                    const uint32_t m = SP.meta[rec];
                    Hc.ipix[0] = 4u * SP.ipix[rec] + (uint32_t)(task & 3);
                    Hc.meta[0] = META(3, META_ISEL(m), RAY_ROOT, 0);
                    Hc.cc[0] = SP.cc[rec];
                    Hc.prev[0] = SP.prev[rec];
                    Hc.dpl[0] = SP.dpl[rec];
                }
                Hc.lum[il] = (il < P.L) ? SP.lum[(unsigned long long)rec * LP + il] : 0.0;
            }
            nH = 1; level = 8;
        }
        wave_fence();
        double lum_lost_acc = 0.0;
        for (int it = 0; it < (1 << 20); it++) {
            if (nH > 0) {
                int nHn = 0;
                trace_items<LP>(P, a, S, 0, Hc, nH, Hn, nHn, LW, nL, ST, ostat, level, T, lum_lost_acc, wc, err);
                wave_fence();
                ListView t = Hc; Hc = Hn; Hn = t;
                nH = nHn;
                if (level == a.export_level) {
                    // phase 1 of the sub-tree scheme: hand the blocked nside-4 rays to phase 2
                    int base = 0;
                    if (lane == 0 && nH > 0) base = atomicAdd(a.spawn_counter, nH);
                    base = first_lane_i(base);
                    int* ssrc = (int*)(a.spawn + list_bytes(a.spawn_cap, LP));
                    for (int i = lane; i < nH; i += 64) {
                        const int pos = base + i;
                        if (pos < a.spawn_cap) {
                            SP.ipix[pos] = Hc.ipix[i]; SP.meta[pos] = Hc.meta[i]; SP.cc[pos] = Hc.cc[i];
                            SP.prev[pos] = Hc.prev[i]; SP.dpl[pos] = Hc.dpl[i]; ssrc[pos] = nc;
                            for (int k = 0; k < P.L; k++) SP.lum[(unsigned long long)pos * LP + k] = Hc.lum[(unsigned long long)i * LP + k];
                        } else err |= ERR_CAPACITY;
                    }
                    nH = 0;
                    break;
                }
            } else if (nL > 0) {
                int nE = 0;
                TSTAMP(te0);
                extract_low<LP>(P, a, LW, nL, Hn, nE, Hc, nH, hkey, hmem, ostat, level, err);
                TSTAMP(te1);
                TACC(t_extract, te0, te1);
                wave_fence();
                if (nE > 0) trace_items<LP>(P, a, S, 1, Hn, nE, Hc, nH, LW, nL, ST, ostat, level, T, lum_lost_acc, wc, err);
                wave_fence();
            }
            // define_next_level (rt_routines.f90:2745-2770)
            if (nH > 0) level *= 2;
            else if (nL > 0) level /= 2;
            else break;
            err = wave_or(err);
            if (level > 8192 || level < 4) err |= ERR_NSIDE;
            if (err) break;
        }
        // lum_lost (rt_routines.f90:1444-1451): one atomic per wavelength
        if (a.scatter) {
            double v = lum_lost_acc;
            for (int off = LP; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
            if (lane < LP && lane < P.L && v != 0.0) atomicAdd(&P.lum_lost[lane], v);
        }
        if (a.record_cost && lane == 0) atomicAdd(&P.src_cost[nc], (double)(__builtin_amdgcn_s_memtime() - t_task));
        err = wave_or(err);
        if (err) break;
    }
    // counters: one atomic per wave
    unsigned long long ci = wc.inter, cr = wc.rays, cd = wc.deps, cl = wc.lamdep, cu = wc.uadd, cur = wc.uread;
    for (int off = 1; off < 64; off <<= 1) {
        ci += __shfl_xor(ci, off, 64); cr += __shfl_xor(cr, off, 64); cd += __shfl_xor(cd, off, 64);
        cl += __shfl_xor(cl, off, 64); cu += __shfl_xor(cu, off, 64); cur += __shfl_xor(cur, off, 64);
    }
    if (lane == 0) {
        atomicAdd(&P.counters[CNT_INTERSECTIONS], ci);
        atomicAdd(&P.counters[CNT_RAYS], cr);
        atomicAdd(&P.counters[CNT_DEPOSITS], cd);
        atomicAdd(&P.counters[CNT_SCASPE], wc.sca);
        atomicAdd(&P.counters[CNT_LAMDEP], cl);
        atomicAdd(&P.counters[CNT_UADD], cu);
        atomicAdd(&P.counters[CNT_UREAD], cur);
        atomicAdd(&P.counters[CNT_SCASLOTS], wc.slots);
        atomicAdd(&P.counters[CNT_ATOM], wc.atom);
        atomicAdd(&P.counters[CNT_ATOM_LANES], wc.atom_lanes);
#ifdef DRT_PROFILE
        TSTAMP(tk1);
        atomicAdd(&P.counters[8], wc.t_setup); atomicAdd(&P.counters[9], wc.t_step); atomicAdd(&P.counters[10], wc.t_flush);
        atomicAdd(&P.counters[11], wc.t_park); atomicAdd(&P.counters[12], wc.t_next); atomicAdd(&P.counters[13], wc.t_commit);
        atomicAdd(&P.counters[14], wc.t_extract); atomicAdd(&P.counters[15], tk1 - tk0);
        atomicAdd(&P.counters[16], wc.n_steps); atomicAdd(&P.counters[17], wc.n_busy_lanes);
        atomicAdd(&P.counters[18], wc.n_groups); atomicAdd(&P.counters[19], wc.n_deps);
#endif
        if (err) atomicOr(P.error, err);
    }
}

template <int LP>
__global__ void __launch_bounds__(256, DRT_WAVES_FOR(LP)) rt_stage_kernel(const DevProblem* __restrict__ Pp, StageArgs a)
{
    rt_stage_body<LP>(Pp, a);
}
// One wavelength on a large grid: the tree walk misses the caches more often and a third wave per SIMD pays for its
// spills (85 801 cells: 1465 -> 1205 ms; 21 801 cells: 199 -> 224 ms, so small grids keep two waves).
__global__ void __launch_bounds__(256, 3) rt_stage_kernel_1w3(const DevProblem* __restrict__ Pp, StageArgs a)
{
    rt_stage_body<1>(Pp, a);
}

// ------------------------------------------------------------------------------------------------
// HG normalisation table of one HEALPix level (DevProblem::hgnorm): one thread per (pixel, wavelength), the arithmetic
// of the launch-time evaluation (trace_items, trace_items_w) to the letter.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double hg_norm(const DevProblem& P, int l, double nx, double ny, double nz)
{
    const double* tab = P.tab[l];
    if (!tab) return 1.0;
    const double g = P.g[l];
    const int np = P.npix[l], hp = P.npix_hp[l];
    const double ga = 1.0 + g * g, gb = 2.0 * g;
    double tot = 0;
    for (int p = 0; p < hp; p++) {
        const double ads = tab[p] * nx + tab[np + p] * ny + tab[2 * np + p] * nz;
        tot += pow_m1p5(ga - gb * ads);
    }
    return tot;
}

__global__ void k_hgnorm(const DevProblem* __restrict__ Pp, int k, double* __restrict__ out)
{
    const DevProblem& P = *Pp;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long npixk = 12ll << (2 * k);
    if (idx >= npixk * P.L) return;
    const int pix = (int)(idx / P.L), l = (int)(idx % P.L);
    PixDir pd;
    pix2ang_nest_sc(P, 1 << k, pix, pd);
    out[idx] = hg_norm(P, l, pd.st * pd.cp, pd.st * pd.sp, pd.ct);
}

// ------------------------------------------------------------------------------------------------
// rt_loop_iobs (rt_routines.f90:1470-1603): one ray per (source, observer direction), attenuation
// only, to the model edge. Lane = (ray, wavelength).
// ------------------------------------------------------------------------------------------------
template <int LP>
__global__ void __launch_bounds__(256) rt_iobs_kernel(const DevProblem* __restrict__ Pp, int rt, const int* __restrict__ src_list,
                                                      int nsources)
{
    const DevProblem& P = *Pp;
    constexpr int R = 64 / LP;
    const int lane = lane_id();
    const int slot = lane / LP, il = lane % LP;
    const bool lam_ok = il < P.L;
    const int gslot = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * R + slot;
    const int nslots = gridDim.x * (blockDim.x >> 6) * R;
    const long long nrays = (long long)nsources * P.ndir;
    unsigned long long inter = 0, rays = 0, deps = 0;
    int err = 0;
    for (long long q = gslot; q < nrays; q += nslots) {
        const int nc = src_list[q / P.ndir];
        const int j = (int)(q % P.ndir);
        SrcInfo S;
        S.nc = nc; S.lvl_nc = P.lvl[nc];
        int sid = nc;
        if (rt == RTT_I_OBS_DIR_SRC) {
            const double* ps = P.p_src_pos + 3ull * (unsigned long long)P.src_cell[nc];
            S.sx = ps[0]; S.sy = ps[1]; S.sz = ps[2];
            sid = P.N + P.src_cell[nc];
        } else {
            const double* c = P.ccoord + 3ull * (unsigned long long)nc;
            S.sx = c[0]; S.sy = c[1]; S.sz = c[2];
        }
        S.csize_i = P.csize_arr[S.lvl_nc]; S.area_i = P.carea_arr[S.lvl_nc];
        S.norm = 4 * P.pi * P.csize[nc] * P.csize[nc];
        double src_lum = 0.0;
        if (lam_ok) {
            if (rt == RTT_I_OBS_DIR_SRC) src_lum = P.p_src_lum[(unsigned long long)P.src_cell[nc] * P.L + il];
            else if (rt == RTT_I_OBS_DIR_CELL) src_lum = P.lumcell[(unsigned long long)nc * P.L + il];
            else if (!P.iobs_healpix) src_lum = P.scaspe_tot[il][(unsigned long long)nc * P.nstride[il] + P.npix_hp[il] + j] * P.npix_hp[il];
        }
        const double* sc = P.dir_out_sc + 4ull * j;
        const double sin_dl = sc[0], cos_dl = sc[1], sin_al = sc[2], cos_al = sc[3];
        if (lam_ok && rt == RTT_I_OBS && P.iobs_healpix) {
            // assign_src_lum (rt_routines.f90:3043-3071): the pixel of the wavelength's own HEALPix resolution that holds the direction
            const int kp = P.kp[il];
            const int ip = (kp >= 0) ? ang2pix_nest_z(P, 1 << kp, cos_dl, P.dir_out[P.ndir + j]) : 0;
            src_lum = P.scaspe_tot[il][(unsigned long long)nc * P.nstride[il] + ip] * P.npix_hp[il];
        }
        const double d0 = sin_dl * cos_al, d1 = sin_dl * sin_al, d2 = cos_dl;
        Ray r;
        r.cc = nc; r.prev = 0.0; r.isel_old = 0; r.clvl = S.lvl_nc; r.clvl_old = -1; r.cc_new = -1; r.flag_beam = true;
        r.cx = P.cpack[3ull * nc]; r.cy = P.cpack[3ull * nc + 1]; r.cz = P.cpack[3ull * nc + 2];
        int v;
        v = (int)(d0 / P.glepsilon); r.set_inc(0, max(min(v, 1), -1));
        v = (int)(d1 / P.glepsilon); r.set_inc(1, max(min(v, 1), -1));
        v = (int)(d2 / P.glepsilon); r.set_inc(2, max(min(v, 1), -1));
        r.cpi0 = 1 / d0; r.cpi1 = 1 / d1; r.cpi2 = 1 / d2;

        double I = src_lum / S.norm;
        const bool act = lam_ok && I > 0;
        if (il == 0) rays++;
        bool done = false;
        for (int step = 0; step < (1 << 24) && !done; step++) {
            if (il == 0) inter++;
            double psel; int isel;
            calc_psel(P, S, r, psel, isel);
            double length = psel - r.prev;
            if (fabs(length) < P.len_eps * P.csize_arr[r.clvl]) length = 0;
            if (length < 0) { err |= ERR_NEG_LENGTH; break; }
            if (length != 0.) {
                if (il == 0) deps++;
                if (act) {  // the i_obs branch of deposit: attenuation only (rt_routines.f90:2543-2555)
                    double dens = P.dens[(unsigned long long)r.cc * P.L + il];
                    double tau_c = dens * length;
                    double ex = exp(-tau_c);
                    if (r.cc != nc) I = I * ex;
                    else {
                        if (!(tau_c < P.tau_thin || I == 0)) {  // own cell: homogeneous-emission form (:2409-2413)
                            tau_c = dens * S.csize_i;
                            ex = exp(-tau_c);
                        }
                        if (!(tau_c < P.tau_thin)) I = I / (tau_c) * (1 - ex);
                    }
                }
            }
            r.prev = psel; r.isel_old = isel; r.clvl_old = r.clvl;
            const double* c = P.ccoord + 3ull * (unsigned long long)r.cc;
            const int clvl = r.clvl;
            bool inside = step_axis(P, r.cx, r.inc(0), isel == 1, clvl, psel, r.cpi0, S.sx, c[0]);
            inside = inside && step_axis(P, r.cy, r.inc(1), isel == 2, clvl, psel, r.cpi1, S.sy, c[1]);
            inside = inside && step_axis(P, r.cz, r.inc(2), isel == 3, clvl, psel, r.cpi2, S.sz, c[2]);
            if (!inside) done = true;
            else if (!ccindd_to_cc(P, r.cx, r.cy, r.cz, r.cc, r.clvl)) { err |= ERR_CELL_NOT_FOUND; break; }
        }
        if (done && act) {
            double* dst = P.i_obs + ((unsigned long long)j * (P.N + P.nsrc) + sid) * P.L + il;
            *dst = *dst + I;  // each (sid, j, il) is written by exactly one lane per launch
        }
    }
    for (int off = 1; off < 64; off <<= 1) {
        inter += __shfl_xor(inter, off, 64); rays += __shfl_xor(rays, off, 64); deps += __shfl_xor(deps, off, 64);
    }
    err = wave_or(err);
    if (lane == 0) {
        atomicAdd(&P.counters[CNT_INTERSECTIONS], inter);
        atomicAdd(&P.counters[CNT_RAYS], rays);
        atomicAdd(&P.counters[CNT_DEPOSITS], deps);
        if (err) atomicOr(P.error, err);
    }
}
