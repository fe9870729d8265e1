// dart-ray_amd — the ray-cell stage kernel with lane = ray ("ray-major").
//
// rt_stage_kernel<LP> (rt_kernels.hpp) gives every (ray, wavelength) pair a lane: with the 32 wavelengths
// of a TRUST-I run a wave carries 2 rays, computes each ray's geometry 32 times over, and - because the
// u_fest pruning of deposit (rt_routines.f90:2470-2480) switches most wavelengths of a ray off early -
// leaves ~80 % of its lanes idle in the deposit arithmetic (measured: 12 of 64 lanes deposit per step).
// Here a lane owns a RAY for its whole life, like a thread of the reference does, and the wavelengths are
// a wave-uniform loop over the ray's iq list (set_iq_list :1860-1893, kept as a 64-bit mask): geometry,
// queue handling and the tree walk are paid once per ray, wavelengths nobody in the wave has active are
// skipped, and the 64 rays of a step that cross the same cell still leave the wave as one read-modify-write
// per (cell, wavelength). Ray intensities live in LDS ([L][64] doubles per wave), the HG normalisation of
// every (ray, wavelength) in the wave's arena slice.
//
// Everything that decides (calc_psel, the beam tests, find_cc_new2, queue order, sibling merging) is the
// same code as in rt_kernels.hpp; per-wavelength arithmetic is the same sequence of operations.
#pragma once
#include "rt_kernels.hpp"

#ifndef DRT_REFILL_IDLE
#define DRT_REFILL_IDLE 64  // idle lanes are refilled when at least this many of them wait: whole batches keep the rays of a
                           // batch in phase (siblings cross the same cells in the same step -> more rays per atomic) and cost fewer
                           // set-up passes; measured 32: 1801 ms, 48: 1758, 64: 1738 (TRUST-I, 32 wavelengths, 21 801 cells)
#endif
#ifndef DRT_WIDE_WAVES
#define DRT_WIDE_WAVES 3  // waves per SIMD the register allocator provisions for rt_stage_wide (168 registers; with the ray
                          // intensities in the arena LDS no longer caps the CU at 8 waves). Measured on the bench workload
                          // after the deposit loop had been rolled up: 2 waves + intensities in LDS 6.98 s per step, 2 waves +
                          // arena 7.2 s, 3 waves + arena 6.65 s, 4 waves + arena 7.98 s (128 registers: spills in the loop)
#endif
// Several wavelengths: the unit vectors of tables 0 and 1 live in LDS, once per block ([2][3][64] doubles behind the waves'
// areas; 4.34 -> 4.30 s on the bench step); one wavelength: table 0 in registers (LDS there: 501 -> 511 ms).
struct SlotTabs { double x0, y0, z0; const __attribute__((address_space(3))) double* tabs; };
#define DRT_TAB_WORDS (6 * 64)  // unit vectors of scaspe slot `lane` of slot tables 0 and 1 (where they fit the lanes)


// ---- memory discipline of the wavelength loop ---------------------------------------------------------
// (1) gfx950 counts FLAT instructions (generic pointers) in BOTH vmcnt and lgkmcnt. With flat atomics in flight,
//     every wait for a scalar load or an LDS read (lgkmcnt) waits for the atomics to reach memory. The loop
//     therefore uses GLOBAL instructions (address space 1) for its loads and atomics.
// (2) The compiler must assume an atomic may write *Pp, so every DevProblem field used after an atomic is loaded
//     again (s_load + wait, four times per cell group in the first version of flush_block). What the loop needs
//     is read once per kernel: scalars into HotP, the per-wavelength constants into one lane per wavelength
//     (LaneTab, fetched with readlane: no memory at all).
// LDS pointers carry their address space in the type: as generic pointers they become FLAT accesses wherever the
// compiler cannot infer the space (e.g. at -O1), and a FLAT f64 atomic on an LDS address raises
// HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION.
typedef __attribute__((address_space(3))) double ldouble;
// Ray intensities [L][64] of the wave live in the wave's arena slice (after the HG normalisations): in LDS they take 16 KB
// per wave at 32 wavelengths and cap the CU at 8 waves. -DDRT_I_LDS puts them back into LDS (needs DRT_WIDE_WAVES <= 2
// at 32 wavelengths).
#ifndef DRT_I_LDS
#define DRT_I_GLOBAL
#endif
#ifdef DRT_I_GLOBAL
typedef double* iptr_t;
#define DRT_I_WORDS(L) 0
#define IRD(p, i) gload((p) + (i))
#else
typedef ldouble* iptr_t;
#define DRT_I_WORDS(L) (64 * (L))
#define IRD(p, i) ((p)[i])
#endif
#ifndef DRT_OPQ
#define DRT_OPQ 4  // wavelengths whose operands (dens, aux, intensity: 3 x 64 doubles each) are staged in LDS at a time
#endif
// + [DRT_OPQ][64] u_final sums of the chunk's wavelengths by leader lane, + 64 ints: the leader lanes in rank order
// (one wavelength: a chunk of one)
#define DRT_OPQ_OF(LP) ((LP) == 1 ? 1 : DRT_OPQ)
// + 1 word: the longest queue of the running task (kept here, not in a register that would be live across the step loop)
#define DRT_LDS_WORDS(L, LP) (DRT_I_WORDS(L) + (LP) + 5 * DRT_BST + 3 * 64 * DRT_OPQ_OF(LP) + 3 + 64 * DRT_OPQ_OF(LP) + 32 + 1 + 1)
// waves per SIMD of an instantiation; four for one wavelength (128 registers, 5 KB of LDS per wave) was tried after the register
// diet of round 2: 585 against 503 ms on the bench grid, 6.1 against 4.9 ms on config 1 - the loop still spills at 128
#ifndef DRT_WAVES_LP1
#define DRT_WAVES_LP1 DRT_WIDE_WAVES
#endif
#define DRT_WAVES_OF(LP) ((LP) == 1 ? DRT_WAVES_LP1 : DRT_WIDE_WAVES)
struct HotP {
    int L;
    int u_pad;   // scatter stages: u_final sums ride in the padding slot of the scaspe rows (folded by the host after the launch)
    double tau_thin, cs, en_lim;
    const double *dens, *u_fest_r;
    double *u_fest, *u_final;
#ifdef DRT_DEBUG_FLAGS
    int skip_atomics;
#endif
};
#ifdef DRT_DEBUG_FLAGS
#define DRT_ATOMIC(H, addr, v) do { if ((H).skip_atomics) asm volatile("" ::"v"(addr), "v"(v)); else gatomic_add(addr, v); } while (0)
#else
#define DRT_ATOMIC(H, addr, v) gatomic_add(addr, v)
#endif
struct LaneTab {  // lane l: constants of wavelength l
    double g, ksca;
    int np, ns, tid;          // slots, row stride (np rounded up to 8), slot table
    unsigned long long srow;  // scaspe[l]
};

// deposit (rt_routines.f90:2362-2557) for one wavelength of this lane's ray. br: 1 = the ray is narrower
// than the cell wants (u_fest test, :2455-2490), 2 = deposit u_final + scattered energy, 3 = precalc deposit.
// dens = dens_arr(l, cc) and, for br == 1, u_fest = u_fest_arr(l, cc) are loaded by the caller one wavelength ahead.
__device__ __forceinline__ void deposit_lambda(const HotP& P, double ksca_l, int rt, const SrcInfo& S, int br, int cc, int l, double length,
                                               double dens, double u_fest, double beam_i, double beam_f, double csize_f, double vol_f,
                                               double& I, double& u_add, double& en_sca, double& lost, bool& alive, WideCounters& wc)
{
    const int nc = S.nc;
    double tau_c = dens * length;
    double ex = exp(-tau_c);
    double i_av;
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
    if (br == 1) {
        const double u_beam = i_av * csize_f / P.cs / vol_f * S.area_i * beam_f;
        wc.uread++;
        if (u_beam < P.en_lim * u_fest) {
            I = 0;
            lost = i_av * beam_i * S.csize_i * S.csize_i;
        }
        if (I > 0) alive = true;
        return;
    }
    const double v = i_av * length / P.cs / vol_f * S.area_i * beam_i;
    if (br == 3) {
        // precalculation: u_fest takes the road of u_final - summed per cell over the rays of the step and per chunk of
        // wavelengths (flush_lambda with no scattered energy; HotP::u_final points at u_fest in these stages)
        u_add = v;
        if (v != 0.0) wc.uadd++;
    } else {
        u_add = v;
        if (v != 0.0) wc.uadd++;
        const double k = ksca_l;
        if (cc != nc) {
            if (tau_c > P.tau_thin) en_sca = I * (1.0 - ex) * k * S.area_i * beam_i;
            else en_sca = I * tau_c * k * S.area_i * beam_i;
        } else {
            if (tau_c > P.tau_thin) en_sca = I / (tau_c) * (ex + tau_c - 1) * k * S.area_i * beam_i;
            else en_sca = I * tau_c / 2 * k * S.area_i * beam_i;
        }
    }
    // update ray_intensity (rt_routines.f90:2543-2555)
    if (cc != nc) I = I * ex;
    else if (!(tau_c < P.tau_thin)) I = I / (tau_c) * (1 - ex);
}

// x**(-1.5) with fused multiply-adds (weights only: nothing decides on them). Same accuracy as pow_m1p5 (1.3e-14).
// (Round 3: the cube of the raw root corrected to first order - y0^3 (1 + 1.5 (1 - x y0^2)), six instructions and a chain of five
// instead of seven and seven, 2e-13 - was SLOWER in both instantiations that matter: 4.33 against 4.29 s, 510 against 507 ms.)
__device__ __forceinline__ double pow_m1p5_f(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = y * __builtin_fma(-(hx * y), y, 1.5);
    return (y * y) * y;
}

// process_scatt_rad (rt_routines.f90:2638-2726) + the u_final update (:2497-2505) for ONE wavelength l of all rays
// of this step. M = lanes (rays) with something to deposit; rays in the same cell are summed on chip. Lanes
// 0..np-1 are the scaspe slots of the cell, lane 63 carries u_final in the same atomic instruction.
//
// Slot tables: set_npix_arr (rt_routines.f90:5235-5315) gives a wavelength 12*4^kp HEALPix slots + tot_ndir
// observer slots. Only kp = 0 and kp = 1 can fit the 63 slot lanes, so the unit vectors of (at most) those two
// tables live in registers (SlotTabs, slot = lane) and every wavelength using one of them - and the isotropic
// ones - takes the merged path: ONE f64 atomic instruction per (cell group, wavelength). Tables with more
// slots than lanes (kp >= 2) go through memory 64 slots at a time, merged over the rays of the group as well.
//
// The code exists once (the wavelength loop of the caller is not unrolled): the stage kernel's hot loop stays small
// enough for the instruction cache two CUs share, and one C / U / accumulator pair is live instead of four.
// Cross-lane broadcasts go through LDS: the depositing lanes stage (C, U) of the wavelength and every ray its
// direction (bst: [5][DRT_BST] doubles per wave: C, U, nx, ny, nz); the member loop then reads a ray's values with
// uniform-address ds_reads (one LDS instruction per double, no VALU slot) instead of two v_readlane per double -
// the loop is VALU-issue-bound and ten of its ~22 vector instructions per ray were readlanes.
#define DRT_BST 64  // row length of the broadcast staging area
template <bool BIG, bool TLDS>
__device__ __forceinline__ unsigned long long flush_lambda(const DevProblem& P, unsigned long long M, unsigned long long MC, int cc, double U, const ldouble* bst,
                                             int l, const SlotTabs& T, const HotP& H, const LaneTab& LT, WideCounters& wc,
                                             unsigned long long Mbr = 0ull, ldouble* urow = nullptr)
{
    const int lane = lane_id();
    const unsigned long long Mall = M;
    // wave-uniform constants of the wavelength (scalar registers)
    const double g = readlane_d(LT.g, l);
    const int np = __builtin_amdgcn_readlane(LT.np, l);
    const int ns = __builtin_amdgcn_readlane(LT.ns, l);
    const int tid = __builtin_amdgcn_readlane(LT.tid, l);
    const int reg = (tid < 0) ? -1 : (tid <= 1 && np <= 63) ? tid : 2;  // 0 / 1: register table, -1: isotropic, 2: memory table
    const unsigned long long srow0 = readlane_u64(LT.srow, l);
    const double ga = 1.0 + g * g, gb = 2.0 * g;
    double tx = T.x0, ty = T.y0, tz = T.z0;
    if constexpr (TLDS) {
        const int toff = (reg == 1) ? 3 * 64 : 0;
        tx = T.tabs[toff + lane]; ty = T.tabs[toff + 64 + lane]; tz = T.tabs[toff + 128 + lane];
    }
    const bool slot_lane = lane < np;   // lanes that are scaspe slots of this wavelength (all of them below np <= 63)
    unsigned na = 0, nl = 0;            // atomic instructions and their active lanes (wave-uniform, scalar registers)
    // u_final: the sum of a cell group is handed to the group's first lane; ONE atomic instruction after the loop adds
    // the sums of all groups (each lane to its own cell). The row atomics then have a scalar base and the lane as offset.
    double su_lead = 0.0;
    unsigned long long leaders = 0ull;
    while (M) {
        TSTAMP(tg0);
        const int src = __ffsll((long long)M) - 1;
        const int cc_b = __builtin_amdgcn_readlane(cc, src);
        // the group's u_final sum goes to the first lane of the cell among ALL depositing rays of the step (Mbr): the same
        // lane for every wavelength of the chunk, see the chunk's epilogue in trace_items_w
        const unsigned long long same = __ballot(cc == cc_b);
        const unsigned long long grp = same & Mall;
        const int lead = urow ? __ffsll((long long)(same & Mbr)) - 1 : src;
        M &= ~grp;
        leaders |= 1ull << lead;
#ifdef DRT_PROFILE
        wc.n_groups++; wc.n_memb += __popcll(grp);
        if (reg < 0) wc.n_cls2 += __popcll(grp); else if (reg == 0) wc.n_cls0 += __popcll(grp); else wc.n_cls1 += __popcll(grp);
#endif
        double su = 0.0;
        unsigned long long gc = grp & MC;   // members with scattered energy
        unsigned long long gu = grp & ~MC;  // members that only add to u_final
        while (gu) {  // rays in cells without dust: no LDS round trip, their U is still in a register
            const int m = __ffsll((long long)gu) - 1;
            gu &= gu - 1;
            su += readlane_d(U, m);
        }
        TSTAMP(tg1);
        TACC(t_ghead, tg0, tg1);
        if (!BIG || reg < 2) {
            if (grp & MC) {
                double acc = 0.0;
                if (reg < 0) {  // isotropic: ffn = 1/npix_hp with npix_hp = 1 (:2266)
                    while (gc) {
                        const int m = __ffsll((long long)gc) - 1;
                        gc &= gc - 1;
                        acc += bst[m];
                        su += bst[DRT_BST + m];
                    }
                } else {
                    while (gc) {
                        const int m = __ffsll((long long)gc) - 1;
                        gc &= gc - 1;
                        const double Cm = bst[m];
                        su += bst[DRT_BST + m];
                        const double nx = bst[2 * DRT_BST + m], ny = bst[3 * DRT_BST + m], nz = bst[4 * DRT_BST + m];
                        const double cs = __builtin_fma(tz, nz, __builtin_fma(ty, ny, tx * nx));
                        acc = __builtin_fma(Cm, pow_m1p5_f(__builtin_fma(-gb, cs, ga)), acc);
                    }
                }
                TSTAMP(tg2);
                TACC(t_memb, tg1, tg2);
                double* const rowp = (double*)(srow0 + 8ull * ((unsigned long long)(unsigned)cc_b * (unsigned)ns));  // wave-uniform, 64-byte aligned
                // Several wavelengths: the group's u_final sum rides in the row's padding slot (slot np of a row of ns > np
                // doubles, in the 64-byte line of the row's last slots): no request of its own. rt_loop folds the padding
                // slots into u_final afterwards. (One wavelength: 582 against 514 ms, not done there.)
                bool pad = false;
                if constexpr (TLDS) {
                    pad = H.u_pad && np < ns && np < 64;
                    if (pad) { acc = (lane == np) ? su : acc; su = 0.0; }
                }
                const bool go = (slot_lane || (pad && lane == np)) && acc != 0.0;
                const unsigned long long gb_ = __ballot(go);
                na += (gb_ != 0ull); nl += (unsigned)__popcll(gb_);
                if (go) DRT_ATOMIC(H, rowp + lane, acc);
                TSTAMP(tg3);
                TACC(t_epi, tg2, tg3);
            }
        } else {
            // slot table with more slots than lanes: 64 slots at a time from memory, the rays of the group summed per slot
            const double* tab = P.tab[l];
            double* const rowp = (double*)(srow0 + 8ull * ((unsigned long long)(unsigned)cc_b * (unsigned)ns));
            for (int p0 = 0; p0 < np; p0 += 64) {
                const int p = p0 + lane;
                double px = 0, py = 0, pz = 0;
                if (p < np) { px = gload(tab + p); py = gload(tab + np + p); pz = gload(tab + 2 * np + p); }
                double a2 = 0.0;
                unsigned long long g2 = gc;
                while (g2) {
                    const int m = __ffsll((long long)g2) - 1;
                    g2 &= g2 - 1;
                    const double Cm = bst[m];
                    if (p0 == 0) su += bst[DRT_BST + m];
                    const double nx = bst[2 * DRT_BST + m], ny = bst[3 * DRT_BST + m], nz = bst[4 * DRT_BST + m];
                    const double cs = __builtin_fma(pz, nz, __builtin_fma(py, ny, px * nx));
                    a2 = __builtin_fma(Cm, pow_m1p5_f(__builtin_fma(-gb, cs, ga)), a2);
                }
                const bool go = (p < np) && a2 != 0.0;
                const unsigned long long gb_ = __ballot(go);
                na += (gb_ != 0ull); nl += (unsigned)__popcll(gb_);
                if (go) DRT_ATOMIC(H, rowp + p, a2);
            }
        }
        su_lead = (lane == lead) ? su : su_lead;
    }
    if (urow) {  // u_final leaves the wave once per chunk of wavelengths, not once per wavelength
        if ((leaders >> lane) & 1ull) urow[lane] = su_lead;
        wc.atom += na; wc.atom_lanes += nl;
        return leaders;
    }
    {
        const bool go = ((leaders >> lane) & 1ull) && su_lead != 0.0;
        const unsigned long long gb_ = __ballot(go);
        na += (gb_ != 0ull); nl += (unsigned)__popcll(gb_);
        if (go) DRT_ATOMIC(H, H.u_final + ((unsigned long long)(unsigned)cc * (unsigned)H.L + (unsigned)l), su_lead);
    }
    wc.atom += na; wc.atom_lanes += nl;
    return 0ull;
}

// trace_items of rt_kernels.hpp with lane = ray. Ilds: [L][64] LDS intensities of the wave, thg: [L][64]
// arena, ll: [L] LDS lum_lost of the running task.
// ---- deposit log (MODE 2) -----------------------------------------------------------------------------------------------
// The scattered energy of a step is not spread over the scaspe rows by the tracing kernel: every ray that scatters in its
// cell appends ONE record to a log - key {cell, HEALPix level and pixel of its direction}, the mask of the wavelengths the
// record has room for, and C = en_sca / HG norm of those wavelengths - and k_log_reduce (dartray_rt.hip) later goes through
// the records of each cell, computes the Henyey-Greenstein weights of all its rays and adds every row once. Why: the
// weights are ~12 dependent f64 instructions per (ray, wavelength); inside this kernel they run at three waves per SIMD
// between spills (2.7 s of the 4.2 s bench step, measured by leaving them out), in a kernel of their own at full occupancy
// they take a tenth of that, and the 7.7e10 memory-side atomic requests of the rows (3.8 s of atomic-unit time) are gone.
// Log memory: chunks of StageArgs::log_chunk_bytes taken with one atomic each; a wave fills its chunk record by record
// and writes one index word per record (cell | byte offset / 8 << 27) into the chunk's index region, so that the reduce
// pass can bin the records by cell without parsing them.
struct WaveLog { int cur; unsigned used, nidx; };  // wave-uniform: current chunk (-1: none yet), bytes and index words used in it
#define DRT_LOG_HDR 16  // bytes in front of a record's C values: key, mask
__device__ __forceinline__ unsigned long long log_key(int cell, int nsl, int ipix) { return (unsigned long long)(unsigned)cell | ((unsigned long long)(unsigned)nsl << 27) | ((unsigned long long)(unsigned)ipix << 32); }

template <int LP, int MODE>
__device__ __forceinline__ void trace_items_w(const DevProblem& P, const StageArgs& a, const SrcInfo& S, int mode, const ListView in, int n_in,
                              const ListView outH, int& nH, const ListView outL, int& nL, const ListView stg,
                              int* __restrict__ ostat, int level, iptr_t Ilds, double* __restrict__ thg, ldouble* ll, ldouble* bst, ldouble* opq,
                              const SlotTabs& T, const HotP& H, const LaneTab& LT, WideCounters& wc, int& err, WaveLog& WL)
{
    constexpr bool BIG = (MODE == 1);    // some slot table has more slots than lanes: memory-table path in flush_lambda
    constexpr bool LOGM = (MODE == 2);   // scattered energy goes to the deposit log
    const int lane = lane_id();
    const int L = (LP == 1) ? 1 : H.L;  // one wavelength: the wavelength loops fold away (-4.5 % on the 85 801-cell grid)
    const unsigned long long below = (1ull << lane) - 1ull;
    const int rt = a.rt_type;
    const bool precalc = (rt == RTT_PRECALC_SRC || rt == RTT_PRECALC_CELL);
    const int n_items = (mode == 0) ? 4 * n_in : n_in;
    const int nsl = 31 - __clz(level);
    bool busy = false;
    int my_item = 0;
    int hgpix = 0;                 // row of the HG-normalisation table this ray reads (its pixel; 1 for "reco" rays)
    const double* const hgtab = (nsl <= P.hgnorm_max) ? P.hgnorm[nsl] : nullptr;  // wave-uniform: all rays of this call share nside
    const double beam_i = 4 * P.pi / (12 * (level * level));  // solid angle of a ray of this level (wave-uniform)
    unsigned long long amask = 0;  // the ray's iq list (sticky, set_iq_list :1860-1893)
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
            const int n_busy = __popcll(busy_b);
            const bool more = next_item < chunk_end;
            if (!more && n_busy == 0) break;
            if (++guard > (1ll << 26)) { err |= ERR_STEPS; break; }

            TSTAMP(ts0);
            // ---- refill idle lanes ---------------------------------------------------------------------
            if (more && (n_busy == 0 || (64 - n_busy) >= DRT_REFILL_IDLE)) {
                const unsigned long long idle = ~busy_b;
                const int item = next_item + __popcll(idle & below);
                next_item += __popcll(idle);
                bool fresh = false;
                int parent = 0, cc_old = 0;
                double ct = 1.0, phi = 0.0;
                if (!busy && item < chunk_end) {
                    parent = (mode == 0) ? (item >> 2) : item;
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
                        ct = pd.ct; phi = pd.phi;
                        cc_old = in.cc[parent];
                        const double prev_ray = in.prev[parent];
                        const double d0 = pd.st * pd.cp, d1 = pd.st * pd.sp, d2 = pd.ct;  // calc_cproj :1945
                        bool ok = true;
                        wc.rays++;  // a call of ray_tracing, including those that return at once (:1656)
                        if (P.dbg) {
                            int k = atomicAdd(&P.dbg[0], 1);
                            if (k < P.dbg_cap) {
                                int* d = P.dbg + 1 + 6 * k;
                                d[0] = S.nc; d[1] = level; d[2] = ipix_dir; d[3] = cc_old; d[4] = (type == RAY_ROOT) ? RAY_HIGH : type; d[5] = item;
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
                            int v;  // incvec (rt_routines.f90:1666-1669)
                            v = (int)(d0 / P.glepsilon); r.set_inc(0, max(min(v, 1), -1));
                            v = (int)(d1 / P.glepsilon); r.set_inc(1, max(min(v, 1), -1));
                            v = (int)(d2 / P.glepsilon); r.set_inc(2, max(min(v, 1), -1));
                            r.cpi0 = 1 / d0; r.cpi1 = 1 / d1; r.cpi2 = 1 / d2;
                            double nx = d0, ny = d1, nz = d2;
                            if (a.scatter && is_reco && level <= 256) {  // "reco" rays read table row 1 (calc_ffn_arr)
                                PixDir p1;
                                pix2ang_nest_sc(P, level, 1, p1);
                                nx = p1.st * p1.cp; ny = p1.st * p1.sp; nz = p1.ct;
                            }
                            r.nx = nx; r.ny = ny; r.nz = nz;
                            hgpix = (is_reco && level <= 256) ? 1 : ipix_dir;
                            bst[2 * DRT_BST + lane] = nx; bst[3 * DRT_BST + lane] = ny; bst[4 * DRT_BST + lane] = nz;  // read by every lane in flush_lambda
                            busy = true;
                            fresh = true;
                            my_item = item - chunk_base;
                            amask = 0;
                        }
                    }
                }
                // ---- per wavelength: source luminosity, iq list, HG normalisation (uniform loop) ------
                if (__ballot(fresh)) {
                    int kp_prev = -2, ip = 0;
                    for (int l = 0; l < L; l++) {
                        if (fresh) {
                            double src_lum = in.lum[(unsigned long long)parent * LP + l];
                            if (rt == RTT_SCATT && cc_old == S.nc) {
                                // assign_src_lum (rt_routines.f90:3043-3071)
                                const int kp = P.kp[l];
                                if (kp != kp_prev) { ip = (kp >= 0) ? ang2pix_nest_z(P, 1 << kp, ct, phi) : 0; kp_prev = kp; }
                                src_lum = P.scaspe_prev[l][(unsigned long long)S.nc * P.nstride[l] + ip] * P.npix_hp[l];
                            }
                            const double I = src_lum / S.norm;
                            Ilds[l * 64 + lane] = I;
                            if (I > 0) amask |= 1ull << l;
                        }
                    }
                    if (a.scatter && !hgtab) {
                        // calc_ffn_arr (rt_routines.f90:2186-2281): sum over the HEALPix slots of x**-1.5 - only for rays
                        // finer than the tabulated levels (DevProblem::hgnorm), the others read it when they deposit. The slot
                        // cosines do not depend on the wavelength: wavelengths that share a slot table go through the
                        // slots together, HGB at a time (one dot product per slot instead of one per slot and wavelength).
                        constexpr int HGB = 8;
                        for (int l0 = 0; l0 < L; l0 += HGB) {
                            const unsigned long long want = __ballot(fresh && ((amask >> l0) & ((1ull << HGB) - 1ull)) != 0ull);
                            const double* tab0 = P.tab[l0];
                            bool same = true;
#pragma unroll
                            for (int k = 0; k < HGB; k++) {
                                const int l = min(l0 + k, L - 1);
                                same = same && (P.tab[l] == tab0 || P.tab[l] == nullptr) && (P.tab[l] == nullptr || P.npix_hp[l] == P.npix_hp[l0]);
                            }
                            if (!want) {
                                for (int k = 0; k < HGB && l0 + k < L; k++) if (fresh) thg[(l0 + k) * 64 + lane] = 1.0;
                            } else if (same && tab0) {
                                const int np = P.npix[l0], hp = P.npix_hp[l0];
                                double ga[HGB], gb[HGB], tot[HGB];
#pragma unroll
                                for (int k = 0; k < HGB; k++) {
                                    const double g = P.g[min(l0 + k, L - 1)];
                                    ga[k] = 1.0 + g * g; gb[k] = 2.0 * g; tot[k] = 0.0;
                                }
                                if (fresh) {
                                    for (int p = 0; p < hp; p++) {
                                        const double ads = tab0[p] * r.nx + tab0[np + p] * r.ny + tab0[2 * np + p] * r.nz;
#pragma unroll
                                        for (int k = 0; k < HGB; k++) tot[k] += pow_m1p5(ga[k] - gb[k] * ads);
                                    }
                                }
#pragma unroll
                                for (int k = 0; k < HGB; k++) {
                                    const int l = l0 + k;
                                    if (l < L && fresh) thg[l * 64 + lane] = (P.tab[l] && ((amask >> l) & 1ull)) ? tot[k] : 1.0;
                                }
                            } else {
                                for (int l = l0; l < min(l0 + HGB, L); l++) {
                                    const double* tab = P.tab[l];
                                    double tot = 1.0;
                                    if (tab && fresh && ((amask >> l) & 1ull)) {
                                        const double g = P.g[l];
                                        const int np = P.npix[l], hp = P.npix_hp[l];
                                        const double ga = 1.0 + g * g, gb = 2.0 * g;
                                        tot = 0;
                                        for (int p = 0; p < hp; p++) {
                                            const double ads = tab[p] * r.nx + tab[np + p] * r.ny + tab[2 * np + p] * r.nz;
                                            tot += pow_m1p5(ga - gb * ads);
                                        }
                                    }
                                    if (fresh) thg[l * 64 + lane] = tot;
                                }
                            }
                        }
                    }
                }
            }

            TSTAMP(ts1);
            TACC(t_setup, ts0, ts1);
#ifdef DRT_PROFILE
            wc.n_steps++; wc.n_busy_lanes += __popcll(__ballot(busy));
#endif
            // ---- one ray-cell intersection for every busy lane ---------------------------------------------
            int status = ST_CONT;
            double psel = 0.0, length = 0.0;
            int isel = 0;
            int br = 0;
            double beam_f = 0, csize_f = 0, vol_f = 0;
            // Phase alignment: rays of one source that cross the same cell enter it at nearly the same distance from
            // the source, but after different numbers of steps (different start cells, cell sizes, obliquity). Only
            // the lanes within DRT_PHASE_K cell sizes of the wave's smallest entry distance take this step; the
            // others wait in place, so that the rays of a cell deposit in the same step and leave the wave in one
            // atomic. Pure scheduling: what a ray does does not depend on when it does it. Pays once the kernel is bound by
            // the memory-side atomic unit (bench: 1.99e10 64-byte requests/s = 1.27 TB/s): window 0 -> 6.62 s per step,
            // 1 cell size -> 6.14, 2 -> 5.99, 3 -> 6.39, 4 -> 6.52 (fewer atomics against more, emptier steps).
            bool go = busy;
            if (a.scatter && a.phase_win > 0.0) {
                double pm = busy ? r.prev : 1.0e300;
                for (int off = 1; off < 64; off <<= 1) pm = fmin(pm, __shfl_xor(pm, off, 64));
                go = busy && (r.prev <= pm + a.phase_win * P.csize_arr[r.clvl]);
            }
            // The next cell of every ray that may go on is looked up HERE, in front of the step's atomics: loads and atomics
            // share one in-order counter, so a load issued behind them returns only when they have all been performed. The
            // ray moves there (r.cc, r.clvl) after the deposit, if it goes on at all. (-1 % at one wavelength)
            int n_cc = 0, n_clvl = 0;
            bool n_inside = true, n_found = true;
            if (go) {
                wc.inter++;
                double cc0, cc1, cc2;
                calc_psel(P, S, r, psel, isel, cc0, cc1, cc2);
                {
                    const int cc_keep = r.cc, clvl_keep = r.clvl;
                    n_found = next_cell(P, S, r, isel, psel, cc0, cc1, cc2, n_inside);
                    n_cc = r.cc; n_clvl = r.clvl;
                    r.cc = cc_keep; r.clvl = clvl_keep;
                }
                length = psel - r.prev;
                if (fabs(length) < P.len_eps * P.csize_arr[r.clvl]) length = 0;
                if (length < 0) { err |= ERR_NEG_LENGTH; status = ST_GONE; }
                if (r.cc != S.nc && r.cc != r.cc_new) r.flag_beam = false;
                if (length != 0. && status == ST_CONT) {
                    // deposit, the part that does not depend on the wavelength (rt_routines.f90:2362-2557)
                    wc.deps++;
                    wc.lamdep += (DRT_WIDE_CNT_T)__popcll(amask);
                    const int lv = r.clvl;  // = lvl(cc): set with r.cc wherever the ray enters a cell
                    csize_f = P.csize_arr[lv]; vol_f = P.cvol_arr[lv];
                    beam_f = P.carea_arr[lv] / (psel * psel);
                    const bool narrow = (beam_f / P.bm_par <= beam_i) && !r.flag_beam;
                    const bool fits = (beam_f / P.bm_par_max < beam_i) || (level <= 8) || r.flag_beam;
                    if (precalc) {
                        if (psel > P.rad_lim_ms) status = ST_GONE;
                        else if (narrow) status = ST_GO_HIGH;
                        else if (fits) br = 3;
                        else status = ST_GO_LOW;
                    } else {
                        if (narrow) br = 1;
                        else if (fits) br = 2;
                        else status = ST_GO_LOW;
                    }
                }
            }
            TSTAMP(ts2);
            TACC(t_step, ts1, ts2);
            // ---- the wavelengths of the step: one at a time, operands fetched one wavelength ahead ---------------
            bool alive = false;
            if (__ballot(br != 0)) {
                const int cc_s = r.cc;
                // operands of wavelength l for this lane: dens_arr(l, cc) and, in ONE more load whose meaning follows the
                // branch, u_fest_arr(l, cc) (br 1) or the ray's HG normalisation (br 2)
                const double* const p_dens = H.dens + (unsigned long long)cc_s * L;
                const double* const p_aux = (br == 1) ? H.u_fest_r + (unsigned long long)cc_s * L
                                            : (br == 2) ? (hgtab ? hgtab + (unsigned long long)hgpix * L : thg + lane) : p_dens;
                const int aux_stride = (br == 2 && !hgtab) ? 64 : 1;
                // the wavelengths some ray of this step still carries (wave-uniform): only those are visited
                unsigned long long lam_any;
                if constexpr (LP == 1) lam_any = __ballot(br != 0 && (amask & 1ull)) ? 1ull : 0ull;
                else {
                    unsigned lo = (br != 0) ? (unsigned)amask : 0u, hi = (br != 0) ? (unsigned)(amask >> 32) : 0u;
                    for (int off = 1; off < 64; off <<= 1) lo |= (unsigned)__shfl_xor((int)lo, off, 64);
                    if (LP > 32) for (int off = 1; off < 64; off <<= 1) hi |= (unsigned)__shfl_xor((int)hi, off, 64);
                    lam_any = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)lo)
                              | ((LP > 32) ? ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)hi) << 32) : 0ull);
                }
                // Operands are fetched DRT_OPQ wavelengths at a time into LDS (dens, aux, ray intensity: three loads per
                // wavelength and lane, issued together and waited for once). gfx950 has ONE in-order counter for loads and
                // atomics: a load issued after a wavelength's atomics waits for all of them, so fetching per wavelength
                // drained the atomic queue eleven times per step; per chunk it is two or three.
                constexpr int OPQ = DRT_OPQ_OF(LP);  // one wavelength: nothing to chunk
                // u_final: 256 contiguous bytes per cell at 32 wavelengths. One atomic per wavelength with a lane per cell is
                // one 64-byte request per (cell, wavelength); the sums of a chunk's wavelengths, gathered in LDS by cell and
                // sent as ONE instruction with the lanes (cell, wavelength), are one or two requests per cell - the step runs
                // at the request rate of the memory-side atomic unit.
                constexpr bool UB = (LP > 1) || LOGM;
                ldouble* const ubuf = opq + 3 * 64 * DRT_OPQ_OF(LP);                   // [OPQ][64]
                __attribute__((address_space(3))) int* const lead_lane = (__attribute__((address_space(3))) int*)(ubuf + 64 * DRT_OPQ_OF(LP));
                const unsigned long long Mbr = __ballot(br != 0);
                // log mode: the lead lane of every depositing ray's cell, found once per step (the u_final sums of a cell
                // are gathered in LDS under that lane), and the ray's record of this step
                int mylead = lane, rec_first = -1;
                unsigned long long leaders_step = 0ull;
                double* rec_c = nullptr;
                if constexpr (LOGM) {
                    if (a.scatter || precalc) {
                        unsigned long long rem = Mbr;
                        while (rem) {
                            const int src = __ffsll((long long)rem) - 1;
                            const int cc_b = __builtin_amdgcn_readlane(cc_s, src);
                            const unsigned long long gm = __ballot(cc_s == cc_b) & Mbr;
                            rem &= ~gm;
                            mylead = ((gm >> lane) & 1ull) ? src : mylead;
                            leaders_step |= 1ull << src;
                        }
                    }
                }
                while (lam_any) {
                    unsigned long long chunk_leaders = 0ull;
                    bool cany = false;  // log mode: this ray scatters at some wavelength of the chunk
                    if (UB && (a.scatter || precalc)) {
#pragma unroll
                        for (int q = 0; q < OPQ; q++) ubuf[q * 64 + lane] = 0.0;
                    }
                    int lq[OPQ];
                    unsigned long long mq = lam_any;
#pragma unroll
                    for (int q = 0; q < OPQ; q++) { lq[q] = mq ? __ffsll((long long)mq) - 1 : -1; mq &= mq - 1; }
                    double tdn[OPQ], tax[OPQ], tI[OPQ];
#pragma unroll
                    for (int q = 0; q < OPQ; q++) {
                        tdn[q] = 0.0; tax[q] = 1.0; tI[q] = 0.0;
                        if (lq[q] >= 0 && br != 0 && ((amask >> lq[q]) & 1ull)) {
                            tdn[q] = gload(p_dens + lq[q]); tax[q] = gload(p_aux + lq[q] * aux_stride); tI[q] = IRD(Ilds, lq[q] * 64 + lane);
                        }
                    }
#pragma unroll
                    for (int q = 0; q < OPQ; q++) { opq[(3 * q) * 64 + lane] = tdn[q]; opq[(3 * q + 1) * 64 + lane] = tax[q]; opq[(3 * q + 2) * 64 + lane] = tI[q]; }
#pragma nounroll
                    for (int q = 0; q < OPQ && lam_any; q++) {
                        const int l = __ffsll((long long)lam_any) - 1;
                        lam_any &= lam_any - 1;
                        const bool act = br != 0 && ((amask >> l) & 1ull);
                        if (__ballot(act)) {
                            TSTAMP(td0);
#ifdef DRT_PROFILE
                            wc.n_lam++;
#endif
                            double C = 0.0, U = 0.0;
                            if (act) {
                                const double dn = opq[(3 * q) * 64 + lane], ax = opq[(3 * q + 1) * 64 + lane];
                                double I = opq[(3 * q + 2) * 64 + lane];
                                double e = 0.0, lost = 0.0;
                                deposit_lambda(H, readlane_d(LT.ksca, l), rt, S, br, cc_s, l, length, dn, ax, beam_i, beam_f, csize_f, vol_f, I, U, e, lost, alive, wc);
                                Ilds[l * 64 + lane] = I;
                                if (lost != 0.0) (void)__hip_atomic_fetch_add(ll + l, lost, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                if (e != 0.0) {
                                    C = e / ax;
                                    wc.sca++;
                                    wc.slots += (DRT_WIDE_CNT_T)__builtin_amdgcn_readlane(LT.np, l);
                                }
                            }
                            TSTAMP(td1);
                            TACC(t_dep, td0, td1);
                            if constexpr (LOGM) {
                                // u_final: summed per cell in LDS under the cell's lead lane; C waits in the ray's own LDS slot
                                // (the operand it replaces has been read) for the record written at the end of the chunk
#ifdef DRT_DEBUG_FLAGS
                                if (!(rt == RTT_SCATT && (P.dbg_flags & 16)))   // timing variant: no u_final sums
#endif
                                if (U != 0.0) (void)__hip_atomic_fetch_add(ubuf + q * 64 + mylead, U, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                opq[(3 * q) * 64 + lane] = C;
                                cany = cany || (C != 0.0);
                            } else if (a.scatter || precalc) {
                                const unsigned long long MC = __ballot(C != 0.0);
                                const unsigned long long M = MC | __ballot(U != 0.0);
                                if (M) {
                                    // stage (C, U) of this wavelength for the uniform-address reads of flush_lambda; LDS
                                    // operations of one wave execute in issue order, the fences keep the compiler from
                                    // moving the reads of other lanes' values above these writes
                                    bst[lane] = C; bst[DRT_BST + lane] = U;
                                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                                    __builtin_amdgcn_wave_barrier();
                                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                                    chunk_leaders |= flush_lambda<BIG, (LP > 1)>(P, M, MC, cc_s, U, bst, l, T, H, LT, wc, Mbr, UB ? ubuf + q * 64 : nullptr);
                                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                                    __builtin_amdgcn_wave_barrier();
                                }
                            }
                        } else if constexpr (LOGM) {
                            opq[(3 * q) * 64 + lane] = 0.0;
                        }
                    }
                    if constexpr (LOGM) {
                        chunk_leaders = leaders_step;
                        // records: a ray gets its record in the first chunk in which it scatters, with room for every wavelength
                        // it carries from that chunk's first one on (rays in cells without dust never get one)
#ifdef DRT_DEBUG_FLAGS
                        const bool need = a.scatter && cany && rec_first < 0 && !(rt == RTT_SCATT && (P.dbg_flags & 4));   // timing variant: no records
#else
                        const bool need = a.scatter && cany && rec_first < 0;
#endif
                        const unsigned long long bneed = __ballot(need);
                        if (bneed) {
                            const int lf = lq[0];
                            const unsigned long long rmask = amask & ~((1ull << lf) - 1ull);
                            const unsigned sz = need ? (unsigned)(DRT_LOG_HDR + 8 * __popcll(rmask)) : 0u;
                            unsigned incl = sz;
                            for (int off = 1; off < 64; off <<= 1) { const unsigned t = (unsigned)__shfl_up((int)incl, off, 64); if (lane >= off) incl += t; }
                            const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
                            const unsigned nrec = (unsigned)__popcll(bneed);
                            if (WL.cur < 0 || WL.used + total > a.log_chunk_bytes || WL.nidx + nrec > a.log_chunk_idx) {
                                // next chunk: the old one's index count is published, a new one is taken
                                int k = 0;
                                if (lane == 0) {
                                    if (WL.cur >= 0) a.log_chunk_count[WL.cur] = WL.nidx;
                                    k = atomicAdd(a.log_chunks_used, 1);
                                }
                                k = first_lane_i(k);
                                if (k >= a.log_nchunks) { err |= ERR_LOG; WL.cur = -1; }
                                else { WL.cur = k; WL.used = 0u; WL.nidx = 0u; }
                            }
                            if (WL.cur >= 0 && need) {
                                const unsigned long long boff = (unsigned long long)(unsigned)WL.cur * a.log_chunk_bytes + WL.used + (incl - sz);
                                unsigned long long* const rec = (unsigned long long*)(a.log + boff);
                                rec[0] = log_key(cc_s, nsl, hgpix);
                                rec[1] = rmask;
                                rec_c = (double*)(rec + 2);
                                rec_first = lf;
                                a.log_idx[(unsigned long long)(unsigned)WL.cur * a.log_chunk_idx + WL.nidx + (unsigned)__popcll(bneed & lanemask_lt())] = (unsigned long long)(unsigned)cc_s | ((boff >> 3) << 27);
                                atomicAdd(a.log_cell_count + cc_s, 1u);   // records per cell, for the reduce pass's binning (no return value: the atomic unit has room)
                            }
                            if (WL.cur >= 0) { WL.used += total; WL.nidx += nrec; }
                        }
                        if (rec_first >= 0) {
                            const unsigned long long rmask = amask & ~((1ull << rec_first) - 1ull);
#pragma unroll
                            for (int q = 0; q < OPQ; q++) {
                                const int l = lq[q];
#ifdef DRT_DEBUG_FLAGS
                                if (rt == RTT_SCATT && (P.dbg_flags & 8)) continue;   // timing variant: records allocated, C not stored
#endif
                                if (l >= 0 && ((rmask >> l) & 1ull)) rec_c[__popcll(rmask & ((1ull << l) - 1ull))] = opq[(3 * q) * 64 + lane];
                            }
                        }
                    }
                    if (UB && chunk_leaders) {
                        // epilogue of the chunk: lane p = 4 * (rank of a leader) + (slot of the chunk) adds that leader's sum
                        // of that wavelength to u_final(l, cell of the leader)
                        if ((chunk_leaders >> lane) & 1ull) lead_lane[__popcll(chunk_leaders & lanemask_lt())] = lane;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        const int nlead = __popcll(chunk_leaders);
                        unsigned na = 0, nl = 0;
                        for (int p0 = 0; p0 < OPQ * nlead; p0 += 64) {
                            const int p = p0 + lane, k = p / OPQ, q = p % OPQ;
                            const bool in = k < nlead;
                            const int f = in ? lead_lane[k] : 0;
                            const int c_f = __shfl(cc_s, f, 64);
                            double v = 0.0;
                            int l_q = lq[0];
#pragma unroll
                            for (int j = 1; j < OPQ; j++) l_q = (q == j) ? lq[j] : l_q;
                            if (in) v = ubuf[q * 64 + f];
                            const bool go_u = in && v != 0.0 && l_q >= 0;
                            const unsigned long long gb_ = __ballot(go_u);
                            na += (gb_ != 0ull); nl += (unsigned)__popcll(gb_);
                            if (go_u) DRT_ATOMIC(H, H.u_final + ((unsigned long long)(unsigned)c_f * (unsigned)H.L + (unsigned)l_q), v);
                        }
                        wc.atom += na; wc.atom_lanes += nl;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
            if (br == 1) {
                const bool wh = wall_hit(P, rt, r.cc);
                status = (alive && !wh) ? ST_GO_HIGH : ST_GONE;
            }
            if (a.psel_sum) {
                // psel_av (rt_routines.f90:1746-1748, 1792-1793): the path of a ray that ends here - given up (ras_gone) or
                // leaving the model - is added to the task's sum (one LDS word per wave, no register)
                const bool ended = go && (status == ST_GONE || (status == ST_CONT && n_found && !n_inside));
                const unsigned long long be = __ballot(ended);
                if (be) {
                    wc.psel_n += (unsigned)__popcll(be);
                    if (ended) (void)__hip_atomic_fetch_add(opq + 3 * 64 * DRT_OPQ_OF(LP) + 64 * DRT_OPQ_OF(LP) + 32, psel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            TSTAMP(ts3);
            TACC(t_flush, ts2, ts3);

            // ---- blocked rays are parked at their item's index (add_to_high/low_ray_list, ray_list.f90:463-579)
            if (go && (status == ST_GO_HIGH || status == ST_GO_LOW)) {
                const bool hi = (status == ST_GO_HIGH);
                const double dpl = r.prev / ((r.isel_old == 1) ? r.cpi0 : (r.isel_old == 2) ? r.cpi1 : r.cpi2);
                stg.ipix[my_item] = (uint32_t)r.ipix_dir;
                stg.meta[my_item] = META(nsl, r.isel_old, hi ? RAY_HIGH : RAY_LOW, 0);
                stg.cc[my_item] = r.cc;
                // add_to_high zeroes prev/dplane when isel_old == 0 (ray_list.f90:496-503)
                stg.prev[my_item] = (hi && r.isel_old == 0) ? 0.0 : r.prev;
                stg.dpl[my_item] = (hi && r.isel_old == 0) ? 0.0 : dpl;
                ostat[my_item] = hi ? 1 : 2;
                for (int l = 0; l < L; l++) stg.lum[(unsigned long long)my_item * LP + l] = Ilds[l * 64 + lane] * S.norm;
            }
            if (go && status != ST_CONT) { busy = false; go = false; }
            TSTAMP(ts4);
            TACC(t_park, ts3, ts4);

            // ---- next cell --------------------------------------------------------------------------------
            if (go) {
                r.prev = psel;
                r.isel_old = isel;
                r.clvl_old = r.clvl;
                r.cc = n_cc; r.clvl = n_clvl;
                if (!n_found) { err |= ERR_CELL_NOT_FOUND; busy = false; }
                if (!n_inside) busy = false;
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
                    for (int k = 0; k < L; k++) o.lum[(unsigned long long)pos * LP + k] = stg.lum[(unsigned long long)i * LP + k];
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

// rt_loop + main_dir_loop (rt_routines.f90:985-1058, 1343-1467), ray-major. LP is the row stride of the
// luminosity arrays in the queues (next power of two >= lnum); BIG: some slot table has more slots than lanes (kp >= 2),
// the instantiation without it has no memory-table path in its deposit loop.
// Dynamic LDS: per wave DRT_LDS_WORDS(L, LP) doubles.
template <int LP, int MODE>
__device__ __forceinline__ void rt_stage_wide_body(const DevProblem* __restrict__ Pp, const StageArgs& a)
{
    constexpr bool LOGM = (MODE == 2);
    extern __shared__ double lds_w[];
    const DevProblem& P = *Pp;
    const int lane = lane_id();
    const int L = (LP == 1) ? 1 : P.L;
    const int waves_per_block = blockDim.x >> 6;
    // wave-uniform by construction, and told so: everything carved from the arena below (queue views, hash, HG table)
    // then lives in scalar registers instead of 50+ vector registers that are spilled inside divergent code
    const int wib = first_lane_i((int)(threadIdx.x >> 6));
    const int gwave = blockIdx.x * waves_per_block + wib;
    ldouble* ll = (ldouble*)lds_w + wib * DRT_LDS_WORDS(L, LP) + DRT_I_WORDS(L);
    ldouble* bst = ll + LP;  // [5][DRT_BST]: C, U of the wavelength being flushed; nx, ny, nz of every ray;
    ldouble* opq = bst + 5 * DRT_BST;  // [DRT_OPQ][3][64]: dens, aux, intensity of the wavelengths being deposited 
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    lds_u32* const qw = (lds_u32*)(ll + (DRT_LDS_WORDS(L, LP) - DRT_I_WORDS(L) - 1));   // the last word of the wave's area
    char* abase = P.arena + (unsigned long long)gwave * a.wave_stride;
    const unsigned long long lbytes = list_bytes(a.cap, LP);
    const ArenaLayout AL = arena_layout(a.cap, a.hcap, LP);
    int* hkey = (int*)(abase + AL.hkey);
    int* hmem = (int*)(abase + AL.hmem);
    int* ostat = (int*)(abase + AL.ostat);
    double* thg = (double*)(abase + AL.thg);  // [L][64]
#ifdef DRT_I_GLOBAL
    iptr_t Ilds = (double*)(abase + AL.inten);  // [L][64]
#else
    iptr_t Ilds = ll - DRT_I_WORDS(L);
#endif
    const int G = a.sectors_per_task;
    const int tasks_per_src = 48 / G;
    const bool from_spawn = a.spawn_count > 0;
    const int ntasks = from_spawn ? 4 * a.spawn_count : a.n_heavy + a.nsources * tasks_per_src;
    const ListView SP = list_view(a.spawn, 0, a.spawn_cap, LP, 0);
    const int* spawn_src = (const int*)(a.spawn + list_bytes(a.spawn_cap, LP));
    const int rt = a.rt_type;
    WideCounters wc = {};
    int err = 0;
    WaveLog WL = {-1, 0u, 0u};
    TSTAMP(tk0);
    HotP H;
    H.u_pad = (a.scatter && !LOGM) ? 1 : 0;  // log mode: no row atomics for u_final to ride in
    H.L = L; H.tau_thin = P.tau_thin; H.cs = P.cs; H.en_lim = P.en_lim;
    H.dens = P.dens; H.u_fest_r = P.u_fest; H.u_fest = P.u_fest; H.u_final = P.u_final;
    if (rt == RTT_PRECALC_SRC || rt == RTT_PRECALC_CELL) H.u_final = P.u_fest;  // the accumulator of these stages
#ifdef DRT_DEBUG_FLAGS
    H.skip_atomics = P.dbg_flags & 1;
#endif
    LaneTab LT = {0.0, 0.0, 0, 0, -1, 0ull};
    if constexpr (LP == 1) {  // one wavelength: the row of the table is the same in every lane and lives in scalar registers
        LT.g = P.g[0]; LT.ksca = P.ksca[0]; LT.np = P.npix[0]; LT.ns = P.nstride[0]; LT.tid = P.tabid[0]; LT.srow = (unsigned long long)P.scaspe[0];
    } else if (lane < P.L) { LT.g = P.g[lane]; LT.ksca = P.ksca[lane]; LT.np = P.npix[lane]; LT.ns = P.nstride[lane]; LT.tid = P.tabid[lane]; LT.srow = (unsigned long long)P.scaspe[lane]; }
    SlotTabs T = {0, 0, 0, nullptr};
    if constexpr (LP == 1) {
        if (a.scatter && P.utab[0] && P.utab_np[0] <= 63 && lane < P.utab_np[0]) { const int n = P.utab_np[0]; T.x0 = P.utab[0][lane]; T.y0 = P.utab[0][n + lane]; T.z0 = P.utab[0][2 * n + lane]; }
    } else {
        ldouble* const tabs_w = (ldouble*)lds_w + waves_per_block * DRT_LDS_WORDS(L, LP);
        T.tabs = tabs_w;
        if (wib == 0) {
            for (int q = 0; q < 6; q++) tabs_w[q * 64 + lane] = 0.0;
            if (a.scatter) {
                if (P.utab[0] && P.utab_np[0] <= 63 && lane < P.utab_np[0]) { const int n = P.utab_np[0]; tabs_w[lane] = P.utab[0][lane]; tabs_w[64 + lane] = P.utab[0][n + lane]; tabs_w[128 + lane] = P.utab[0][2 * n + lane]; }
                if (P.utab[1] && P.utab_np[1] <= 63 && lane < P.utab_np[1]) { const int n = P.utab_np[1]; tabs_w[192 + lane] = P.utab[1][lane]; tabs_w[256 + lane] = P.utab[1][n + lane]; tabs_w[320 + lane] = P.utab[1][2 * n + lane]; }
            }
        }
        __syncthreads();
    }
    if (lane < LP) ll[lane] = 0.0;
    if (lane == 0) opq[3 * 64 * DRT_OPQ_OF(LP) + 64 * DRT_OPQ_OF(LP) + 32] = 0.0;   // the task's sum of ended rays' paths (psel_av)
    // the per-lane tallies: one atomic per wave and counter
    auto flush_counters = [&]() {
        unsigned long long c8[8] = {wc.inter, wc.rays, wc.deps, wc.sca, wc.lamdep, wc.uadd, wc.uread, wc.slots};
        for (int off = 1; off < 64; off <<= 1)
            for (int q = 0; q < 8; q++) c8[q] += __shfl_xor(c8[q], off, 64);
        if (lane == 0) {
            atomicAdd(&P.counters[CNT_INTERSECTIONS], c8[0]);
            atomicAdd(&P.counters[CNT_RAYS], c8[1]);
            atomicAdd(&P.counters[CNT_DEPOSITS], c8[2]);
            atomicAdd(&P.counters[CNT_SCASPE], c8[3]);
            atomicAdd(&P.counters[CNT_LAMDEP], c8[4]);
            atomicAdd(&P.counters[CNT_UADD], c8[5]);
            atomicAdd(&P.counters[CNT_UREAD], c8[6]);
            atomicAdd(&P.counters[CNT_SCASLOTS], c8[7]);
        }
        wc.inter = wc.rays = wc.deps = wc.sca = wc.lamdep = wc.uadd = wc.uread = wc.slots = 0;
    };

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
        const bool psrc = (rt == RTT_PRECALC_SRC || rt == RTT_DIR_SRC);
        if (psrc) {  // calc_src_ccoord (rt_routines.f90:1922-1939)
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
        int nH = Gt, nL = 0, level = 4;
        if (lane == 0) *qw = 0u;   // the longest high / low queue of this task
        if (!from_spawn) {
            if (lane < Gt) {
                Hc.ipix[lane] = (uint32_t)(sec0 + lane);  // idir*4 + idir2 at nside 2
                Hc.meta[lane] = META(1, 0, RAY_HIGH, 0);
                Hc.cc[lane] = nc;
                Hc.prev[lane] = 0.0;
                Hc.dpl[lane] = 0.0;
                for (int l = 0; l < L; l++)
                    Hc.lum[(unsigned long long)lane * LP + l] = psrc ? P.p_src_lum[(unsigned long long)P.src_cell[nc] * L + l] : P.lumcell[(unsigned long long)nc * L + l];
            }
        } else {
            // sub-tree task: child (task & 3) of the exported nside-4 ray, launched at nside 8
            const int rec = task >> 2;
            if (lane == 0) {
                const uint32_t m = SP.meta[rec];
                Hc.ipix[0] = 4u * SP.ipix[rec] + (uint32_t)(task & 3);
                Hc.meta[0] = META(3, META_ISEL(m), RAY_ROOT, 0);
                Hc.cc[0] = SP.cc[rec];
                Hc.prev[0] = SP.prev[rec];
                Hc.dpl[0] = SP.dpl[rec];
            }
            if (lane < LP) Hc.lum[lane] = (lane < L) ? SP.lum[(unsigned long long)rec * LP + lane] : 0.0;
            nH = 1; level = 8;
        }
        wave_fence();
        for (int it = 0; it < (1 << 20); it++) {
            // one call site for both kinds of pass: mode 0 traces the children of the high queue into the
            // other high queue, mode 1 the merged rays extract_low left in Hn (its "reco" rays went to Hc)
            const int mode = (nH > 0) ? 0 : 1;
            int n_in = nH, n_oh = 0;
            if (mode == 1) {
                int nE = 0;
                TSTAMP(te0);
                extract_low<LP>(P, a, LW, nL, Hn, nE, Hc, nH, hkey, hmem, ostat, level, err);
                TSTAMP(te1);
                TACC(t_extract, te0, te1);
                wave_fence();
                n_in = nE; n_oh = nH;
            }
            const ListView in = (mode == 0) ? Hc : Hn;
            const ListView oh = (mode == 0) ? Hn : Hc;
            if (n_in > 0) trace_items_w<LP, MODE>(P, a, S, mode, in, n_in, oh, n_oh, LW, nL, ST, ostat, level, Ilds, thg, ll, bst, opq, T, H, LT, wc, err, WL);
            wave_fence();
            if (mode == 0) {
                ListView t = Hc; Hc = Hn; Hn = t;
                if (level == a.export_level) {
                    // phase 1 of the sub-tree scheme: hand the blocked nside-4 rays to phase 2
                    nH = n_oh;
                    int base = 0;
                    if (lane == 0 && nH > 0) base = atomicAdd(a.spawn_counter, nH);
                    base = first_lane_i(base);
                    int* ssrc = (int*)(a.spawn + list_bytes(a.spawn_cap, LP));
                    for (int i = lane; i < nH; i += 64) {
                        const int pos = base + i;
                        if (pos < a.spawn_cap) {
                            SP.ipix[pos] = Hc.ipix[i]; SP.meta[pos] = Hc.meta[i]; SP.cc[pos] = Hc.cc[i];
                            SP.prev[pos] = Hc.prev[i]; SP.dpl[pos] = Hc.dpl[i]; ssrc[pos] = nc;
                            for (int k = 0; k < L; k++) SP.lum[(unsigned long long)pos * LP + k] = Hc.lum[(unsigned long long)i * LP + k];
                        } else err |= ERR_CAPACITY;
                    }
                    nH = 0;
                    break;
                }
            }
            nH = n_oh;
            if (lane == 0) (void)__hip_atomic_fetch_max(qw, (unsigned)max(nH, nL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
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
            wave_fence();
            if (lane < L) {
                const double v = ll[lane];
                if (v != 0.0) { atomicAdd(&P.lum_lost[lane], v); ll[lane] = 0.0; }
            }
            wave_fence();
        }
        if (a.psel_sum) {
            wave_fence();
            ldouble* const ps = opq + 3 * 64 * DRT_OPQ_OF(LP) + 64 * DRT_OPQ_OF(LP) + 32;
            if (lane == 0 && wc.psel_n) {
                const int sid = (rt == RTT_PRECALC_SRC || rt == RTT_DIR_SRC) ? P.N + P.src_cell[nc] : nc;
                atomicAdd(&a.psel_sum[sid], (double)*ps);
                atomicAdd(&a.psel_cnt[sid], (double)wc.psel_n);
                *ps = 0.0;
            }
            wc.psel_n = 0;
            wave_fence();
        }
        if (a.record_cost && lane == 0) {
            atomicAdd(&P.src_cost[nc], (double)(__builtin_amdgcn_s_memtime() - t_task));
            // the longest queue among the source's tasks (the order of non-negative doubles is the order of their bit patterns).
            // The host scales it to 48 sectors per task - it knows how many the source's tasks had - and drops it for the sources
            // it ran as sub-tree tasks. A task that outgrew its queues (the loop above left with the error in every lane) leaves
            // a bound instead - three times the capacity, so that the repeated launch cuts it four times finer - read by the host
            // before it undoes the launch. (No condition on the kind of task here: keeping one alive across the step loop cost
            // five more scratch loads per step at one wavelength, 558 against 507 ms.)
            atomicMax((unsigned long long*)&P.src_cost[(unsigned long long)P.N + nc], (unsigned long long)__double_as_longlong((double)((err & ERR_CAPACITY) ? 3u * (unsigned)a.cap : *qw)));
        }
        if (sizeof(DRT_WIDE_CNT_T) < 8 && __ballot(((wc.inter | wc.lamdep | wc.slots) >> 30) != 0)) flush_counters();
        err = wave_or(err);
        if (err) break;
    }
    flush_counters();
    if constexpr (LOGM) { if (lane == 0 && WL.cur >= 0) a.log_chunk_count[WL.cur] = WL.nidx; }
    if (lane == 0) {
        atomicAdd(&P.counters[CNT_ATOM], wc.atom);
        atomicAdd(&P.counters[CNT_ATOM_LANES], wc.atom_lanes);
#ifdef DRT_PROFILE
        TSTAMP(tk1);
        atomicAdd(&P.counters[8], wc.t_setup); atomicAdd(&P.counters[9], wc.t_step); atomicAdd(&P.counters[10], wc.t_flush);
        atomicAdd(&P.counters[11], wc.t_park); atomicAdd(&P.counters[12], wc.t_next); atomicAdd(&P.counters[13], wc.t_commit);
        atomicAdd(&P.counters[14], wc.t_extract); atomicAdd(&P.counters[15], tk1 - tk0);
        atomicAdd(&P.counters[16], wc.n_steps); atomicAdd(&P.counters[17], wc.n_busy_lanes);
        atomicAdd(&P.counters[18], wc.n_groups); atomicAdd(&P.counters[19], wc.n_memb);
        atomicAdd(&P.counters[22], wc.t_dep); atomicAdd(&P.counters[23], wc.t_ghead); atomicAdd(&P.counters[24], wc.t_memb);
        atomicAdd(&P.counters[25], wc.t_epi); atomicAdd(&P.counters[26], wc.n_lam);
        atomicAdd(&P.counters[27], wc.n_cls0); atomicAdd(&P.counters[28], wc.n_cls1); atomicAdd(&P.counters[29], wc.n_cls2);
#endif
        if (err) atomicOr(P.error, err);
    }
}

template <int LP, int MODE>
__global__ void __launch_bounds__(256, DRT_WAVES_OF(LP)) rt_stage_wide(const DevProblem* __restrict__ Pp, StageArgs a)
{
    rt_stage_wide_body<LP, MODE>(Pp, a);
}
// One wavelength, cheap sources (TRUST-I tau = 0.01: ~9e2 intersections per source): a step is a chain of memory round trips and
// few lanes are busy; with two waves per SIMD (256 registers) the kernel does not spill and a step is shorter - config 1:
// 3.69 -> 3.04 ms per step. Where sources are expensive the third wave wins (85 801 cells: 510 ms against 625). The host picks
// by the work per source of the stage's previous call (drt_ctx::cheap_sources).
template <int MODE>
__global__ void __launch_bounds__(256, 2) rt_stage_wide_lp1w2(const DevProblem* __restrict__ Pp, StageArgs a)
{
    rt_stage_wide_body<1, MODE>(Pp, a);
}
