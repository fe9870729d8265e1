// dartray_rt.hip — C-ABI library of the MI355X-native DART-Ray ray-cell path (include/dartray_rt.h).
//
// Host side: owns the device copies of the grid and field arrays (reference layout, see DESIGN.md
// section 2), restates the O(N) set-up helpers of rt_prepare on the host (set_en_lim, prepare_p_src,
// calc_total_luminosity, set_npix_arr, create_scaspe), launches the kernels of rt_kernels.hpp and
// performs the stage-end reductions of rt_mpi. No CPU fallback exists: every compute entry point
// needs a gfx950 device.
#include "../../include/dartray_rt.h"
#include "rt_wide.hpp"
#include "rt_log.hpp"

#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <thread>
#include <string>
#include <vector>

// ------------------------------------------------------------------------------------------------
// small device kernels for the stage glue
// ------------------------------------------------------------------------------------------------
// calc_total_luminosity_sca (rt_routines.f90:698-708): lumcell(il,i) = sum(scaspe(0:npix_hp-1,i)) for leaves;
// totals per wavelength for the convergence test (:725-732) and tot_rad_en (:675-684)
__global__ void k_lumcell_sca(const double* __restrict__ scaspe, const int* __restrict__ cchild, double* __restrict__ lumcell,
                              double* __restrict__ tot, int N, int L, int il, int npix, int nhp, int all_pix, int nstride)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    double s = 0.0;
    if (i < N) {
        const double* row = scaspe + (unsigned long long)i * nstride;
        int n = all_pix ? npix : nhp;
        for (int p = 0; p < n; p++) s += row[p];
        lumcell[(unsigned long long)i * L + il] = (cchild[i] == -1) ? s : 0.0;
    }
    // block reduction -> one atomic per wave
    double v = s;
    for (int off = 1; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0 && v != 0.0) atomicAdd(tot, v);
}

__global__ void k_axpy(double* __restrict__ y, const double* __restrict__ x, unsigned long long n)
{
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) y[i] = y[i] + x[i];
}

// reduce_u_final_arr (io_routines.f90:3743-3757): u_final = max(u_final - u_fest, 0) [all-reduce] + u_fest
__global__ void k_u_sub_clamp(double* __restrict__ u_final, const double* __restrict__ u_fest, unsigned long long n)
{
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) { double d = u_final[i] - u_fest[i]; u_final[i] = d < 0 ? 0.0 : d; }
}
__global__ void k_scale_sub(double* __restrict__ a, const double* __restrict__ b, double f, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = a[i] - f * b[i];
}

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
struct drt_ctx {
    int device = 0, ncu = 256;
    std::string err;
    // grid (host copies for the O(N) helpers)
    int N = 0, max_lvl = 0, base[2] = {0, 0};
    double modelsize = 0;
    std::vector<int32_t> cchild, lvl, src_cell, cell_src;
    std::vector<double> ccoord, csize, csize_arr, carea_arr, cvol_arr;
    std::vector<uint32_t> cpack;
    drt_params par{};
    bool have_par = false, have_grid = false, prepared = false, part2 = false;
    int L = 0, LP = 1;
    bool ray_major = false;
    bool big_tables = false;  // some wavelength's slot table has more than 63 slots (set_npix_arr)
    int G_cell[16] = {0};  // per rt_type: sectors per task that last fitted the queue arena after an overflow (0: none met). Per stage
                           // kind since round 3: the direct light of the N-body grid overflows at 8 sectors, its scattering
                           // iterations fill a tenth of their queues at 4 and were held there by the one global value
    // per rt_type: sectors per task of the stage's last successful cell launch and the share of the queue capacity its longest
    // queue took (from the sources' queue lengths): more sectors are tried only where that leaves room (queues grow with the sectors of a task)
    int G_last[16] = {0};
    double q_fill[16] = {0.0};
    double f_last[16] = {0.0};   // ... and the task limit of that launch as a fraction of a wave's share of the stage (1 / hd in plan())
    int G_want[16] = {0};  // per rt_type: sectors per task suggested by the work per source of the last call of that stage
    bool cheap_sources[16] = {false};  // per rt_type: the last call of that stage saw < 4000 intersections per source (no phase window)
    std::vector<double*> d_bak_scaspe;  // snapshot of the accumulators of a stage, restored when a launch is repeated
    double *d_bak_u = nullptr, *d_bak_lost = nullptr, *d_bak_psel = nullptr;
    size_t bak_psel_n = 0;
    unsigned long long* d_bak_cnt = nullptr;  // lane = ray kernel (rt_wide.hpp) instead of lane = (ray, wavelength)
    std::vector<double> ksca, gsca;
    std::vector<int> npix, npix_hp, kp, ik;
    std::vector<int> nstride;  // device row length of the scaspe arrays (npix rounded up to 8 doubles = one 64-byte line)
    int nuniq = 0;
    std::vector<int> kp_unique, npix_unique;
    int nsrc = 0, ndir = 0;
    std::vector<double> p_src_pos, p_src_lum, dir_out;
    std::vector<double> dens_stars_h, lumcell_h, tot_rad_en;
    double en_lim = 0;
    int bm_par_cur = 0, iterations = 0, cnflag = 0;
    // device
    int *d_cchild = nullptr, *d_lvl = nullptr, *d_src_cell = nullptr, *d_src_list = nullptr;
    uint32_t* d_cpack = nullptr;
    uint32_t* d_nbr = nullptr;
    double *d_ccoord = nullptr, *d_csize = nullptr, *d_dens = nullptr, *d_u_fest = nullptr, *d_u_final = nullptr, *d_lumcell = nullptr;
    double *d_p_src_pos = nullptr, *d_p_src_lum = nullptr, *d_dir_out = nullptr, *d_lum_lost = nullptr, *d_lum_lost_prev = nullptr;
    double *d_i_obs = nullptr, *d_tot = nullptr, *d_src_cost = nullptr;
    // Measured-cost scheduling (build_src_list): src_pred_h[i] = the wave time cell source i is expected to take in the launch
    // being prepared; cost_lum_h[i] = its luminosity (sum over the wavelengths) in the launch d_src_cost was measured in,
    // cost_kind = that launch's rt_type (-1: nothing measured yet)
    std::vector<double> src_pred_h, cost_lum_h;
    std::vector<double> src_q_est_h;   // [N]: an estimate of the same for sources nothing was measured of (fit_queue_prior, after an undone launch)
    std::vector<double> src_q_h;   // [N]: the longest ray queue a 48-sector task of the source would hold (measured on its tasks of the last launch
                                   // of this kind of stage that ran it as whole sectors; 0 = not known)
    int cost_kind = -1;
    bool pred_measured = false;  // src_pred_h of the launch being prepared rests on measured wave times (else on the prior alone)
    int algorithm = 0;  // DRT_RTA_MAIN / DRT_RTA_I_OBS (drt_set_algorithm)
    bool psel_av = false;  // print_psel_av (drt_set_psel_av)
    double *d_psel_sum = nullptr, *d_psel_cnt = nullptr;  // [N + nsrc] each
    bool cost_synced = false;  // np_mpi > 1: d_src_cost has been summed over the ranks since the launch that measured it
    std::vector<double*> d_scaspe, d_scaspe_prev, d_scaspe_tot, d_tab;
    DevProblem* d_prob = nullptr;
    DevProblem hprob{};
    char* d_arena = nullptr;
    size_t arena_bytes = 0;   // 0: not configured - sized at allocation from the wavelengths per ray and the device (default_arena_bytes)
    bool arena_auto = true; int arena_lp = 0;
    int waves = 0, cap = 0, hcap = 0;
    unsigned long long wave_stride = 0;
    unsigned long long* d_counters = nullptr;
    int *d_error = nullptr, *d_task = nullptr, *d_spawn_counter = nullptr;
    char* d_spawn = nullptr;
    int spawn_cap = 0;
    size_t src_list_cap = 0;
    int2* d_heavy = nullptr;
    size_t heavy_cap = 0;
    int* d_deep = nullptr;
    size_t deep_cap = 0;
    bool overflow_met = false;
    // deposit log (rt_wide.hpp MODE 2, rt_log.hpp)
    bool use_log = false;
    size_t log_bytes = 0;            // 0: default
    char* d_log = nullptr;
    unsigned long long *d_log_idx = nullptr, *d_log_sorted = nullptr, *d_log_start = nullptr, *d_log_bsum = nullptr;
    unsigned *d_log_chunk_count = nullptr, *d_log_cell_count = nullptr, *d_log_cursor = nullptr;
    int *d_log_used = nullptr, *d_log_rcounter = nullptr;
    int log_nchunks = 0;
    unsigned log_chunk_bytes = 1u << 20, log_chunk_idx = 1u << 14;
    int log_batches[16] = {0};       // per rt_type: batches the stage's launches were cut into last time (0: not run yet)
    long long log_chunks_last[16] = {0};  // per rt_type: chunks the stage filled last time, all batches together
    int log_peak_last[16] = {0};     // per rt_type: chunks the fullest of those batches filled
    double log_ms = 0;               // last rt_loop: device time of the reduce passes (inside kernel_ms)
    int* d_dbg = nullptr;
    int dbg_cap = 0, dbg_flags = 0;
    double* d_ring_tab[14] = {nullptr}, *d_phi0_tab[14] = {nullptr}, *d_phi1_tab[14] = {nullptr};
    double* d_dir_out_sc = nullptr;
    int tab_max_level = 0;
    double* d_hgnorm[14] = {nullptr};
    int hgnorm_max = 0;
    drt_counters cnt{};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // comm (RCCL, loaded lazily)
    void* rccl = nullptr;
    void* comm = nullptr;
    int nranks = 1, rank = 0;
};

static int fail(drt_ctx* c, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}
#define HIPCHK(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(c, DRT_E_HIP, "%s: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

template <class T> static int dev_upload(drt_ctx* c, T** d, const T* h, size_t n)
{
    if (*d) { (void)hipFree(*d); *d = nullptr; }
    HIPCHK(c, hipMalloc((void**)d, (n ? n : 1) * sizeof(T)));
    if (n && h) HIPCHK(c, hipMemcpy(*d, h, n * sizeof(T), hipMemcpyHostToDevice));
    return DRT_OK;
}
template <class T> static int dev_zero(drt_ctx* c, T** d, size_t n)
{
    if (*d) { (void)hipFree(*d); *d = nullptr; }
    HIPCHK(c, hipMalloc((void**)d, (n ? n : 1) * sizeof(T)));
    HIPCHK(c, hipMemset(*d, 0, (n ? n : 1) * sizeof(T)));
    return DRT_OK;
}

static double ipow(double x, int n)  // x**n as a Fortran runtime evaluates it
{
    double r = 1.0;
    while (n > 0) { if (n & 1) r *= x; x *= x; n >>= 1; }
    return r;
}

extern "C" int drt_create(drt_ctx** out, int device)
{
    if (!out) return DRT_E_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return DRT_E_NODEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return DRT_E_NODEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return DRT_E_NODEVICE;  // kernels are built for gfx950 only
    if (hipSetDevice(device) != hipSuccess) return DRT_E_NODEVICE;
    drt_ctx* c = new drt_ctx();
    c->device = device;
    c->ncu = prop.multiProcessorCount;
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) { delete c; return DRT_E_HIP; }
    if (hipMalloc((void**)&c->d_counters, CNT_N * sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc((void**)&c->d_error, sizeof(int)) != hipSuccess || hipMalloc((void**)&c->d_task, sizeof(int)) != hipSuccess ||
        hipMalloc((void**)&c->d_prob, sizeof(DevProblem)) != hipSuccess || hipMalloc((void**)&c->d_tot, DRT_MAXL * sizeof(double)) != hipSuccess) {
        delete c;
        return DRT_E_HIP;
    }
    (void)hipMemset(c->d_counters, 0, CNT_N * sizeof(unsigned long long));
    (void)hipMemset(c->d_error, 0, sizeof(int));
    *out = c;
    return DRT_OK;
}

static void free_fields(drt_ctx* c)
{
    for (auto p : c->d_bak_scaspe) if (p) (void)hipFree(p);
    c->d_bak_scaspe.clear();
    if (c->d_bak_u) { (void)hipFree(c->d_bak_u); c->d_bak_u = nullptr; }
    for (auto p : c->d_scaspe) if (p) (void)hipFree(p);
    for (auto p : c->d_scaspe_prev) if (p) (void)hipFree(p);
    for (auto p : c->d_scaspe_tot) if (p) (void)hipFree(p);
    for (auto p : c->d_tab) if (p) (void)hipFree(p);
    c->d_scaspe.clear(); c->d_scaspe_prev.clear(); c->d_scaspe_tot.clear(); c->d_tab.clear();
}

extern "C" int drt_destroy(drt_ctx* c)
{
    if (!c) return DRT_E_ARG;
    (void)hipSetDevice(c->device);
    free_fields(c);
    void* ptrs[] = {c->d_nbr, c->d_cchild, c->d_lvl, c->d_src_cell, c->d_src_list, c->d_cpack, c->d_ccoord, c->d_csize, c->d_dens, c->d_u_fest,
                    c->d_u_final, c->d_lumcell, c->d_p_src_pos, c->d_p_src_lum, c->d_dir_out, c->d_lum_lost, c->d_lum_lost_prev,
                    c->d_i_obs, c->d_tot, c->d_src_cost, c->d_heavy, c->d_deep, c->d_psel_sum, c->d_psel_cnt, c->d_prob, c->d_arena, c->d_counters, c->d_error, c->d_task, c->d_dbg, c->d_spawn, c->d_spawn_counter,
                    c->d_bak_lost, c->d_bak_cnt, c->d_bak_psel};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (int k = 0; k < 14; k++) { if (c->d_ring_tab[k]) (void)hipFree(c->d_ring_tab[k]); if (c->d_phi0_tab[k]) (void)hipFree(c->d_phi0_tab[k]); if (c->d_phi1_tab[k]) (void)hipFree(c->d_phi1_tab[k]); }
    if (c->d_dir_out_sc) (void)hipFree(c->d_dir_out_sc);
    for (int k = 0; k < 14; k++) if (c->d_hgnorm[k]) (void)hipFree(c->d_hgnorm[k]);
    {
        void* lp[] = {c->d_log, c->d_log_idx, c->d_log_sorted, c->d_log_start, c->d_log_bsum, c->d_log_chunk_count, c->d_log_cell_count, c->d_log_cursor, c->d_log_used, c->d_log_rcounter};
        for (void* q : lp) if (q) (void)hipFree(q);
    }
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    drt_comm_destroy(c);
    delete c;
    return DRT_OK;
}

extern "C" const char* drt_last_error(const drt_ctx* c) { return c ? c->err.c_str() : "null context"; }

extern "C" int drt_configure(drt_ctx* c, size_t arena_bytes, int waves)
{
    if (!c) return DRT_E_ARG;
    if (arena_bytes) { c->arena_bytes = arena_bytes; c->arena_auto = false; }
    if (waves > 0) c->waves = waves;
    for (int k = 0; k < 16; k++) c->G_cell[k] = 0;
    if (c->d_arena) { (void)hipFree(c->d_arena); c->d_arena = nullptr; }
    return DRT_OK;
}

// ---- module state in ---------------------------------------------------------------------------------
extern "C" int drt_set_grid(drt_ctx* c, int32_t N, int32_t max_lvl, const int32_t base[2], const int32_t* cchild,
                            const int32_t* lvl, const int64_t* cindex, const double* ccoord, const double* csize)
{
    if (!c || N < 2 || !base || !cchild || !lvl || !cindex || !ccoord || !csize) return fail(c, DRT_E_ARG, "drt_set_grid: bad argument");
    if (max_lvl < 1 || max_lvl > DRT_MAXLVL) return fail(c, DRT_E_ARG, "drt_set_grid: max_lvl %d outside 1..%d", max_lvl, DRT_MAXLVL);
    if (base[0] < 2 || base[0] > 15 || base[1] < 2 || base[1] > 15) return fail(c, DRT_E_ARG, "drt_set_grid: base must be within 2..15");
    if (cchild[0] != 1) return fail(c, DRT_E_ARG, "drt_set_grid: the root's children must start at index 1");
    HIPCHK(c, hipSetDevice(c->device));
    c->N = N; c->max_lvl = max_lvl; c->base[0] = base[0]; c->base[1] = base[1];
    c->cchild.assign(cchild, cchild + N);
    c->lvl.assign(lvl, lvl + N);
    c->ccoord.assign(ccoord, ccoord + 3ull * N);
    c->csize.assign(csize, csize + N);
    c->modelsize = csize[0];  // make_csize_arr (io_routines.f90:677)
    // set_base (smooth_grid_routines.f90:622-649)
    long long basediv[2];
    for (int i = 0; i < 2; i++) { int j = 0; long long b3 = (long long)base[i] * base[i] * base[i]; while (!(b3 < (1ll << j))) j++; basediv[i] = 1ll << j; }
    double baseinv[2] = {1.0 / (double)base[0], 1.0 / (double)base[1]};
    // make_csize_arr (io_routines.f90:669-696)
    c->csize_arr.assign(max_lvl + 1, 0.0); c->carea_arr.assign(max_lvl + 1, 0.0); c->cvol_arr.assign(max_lvl + 1, 0.0);
    for (int i = 0; i <= max_lvl; i++) {
        double cs = c->modelsize;
        if (i > 0) cs = c->modelsize * (baseinv[0] * ipow(baseinv[1], i - 1));
        c->csize_arr[i] = cs; c->carea_arr[i] = cs * cs; c->cvol_arr[i] = cs * cs * cs;
    }
    // cindex_to_ccindd (smooth_grid_routines.f90:793-814) for every cell, packed 4 bits per level
    c->cpack.assign(3ull * N, 0u);
    for (int n = 0; n < N; n++) {
        if (lvl[n] < 0 || lvl[n] > max_lvl) return fail(c, DRT_E_ARG, "drt_set_grid: lvl(%d) = %d out of range", n, lvl[n]);
        if (cchild[n] != -1 && (cchild[n] < 1 || cchild[n] >= N)) return fail(c, DRT_E_ARG, "drt_set_grid: cchild(%d) = %d out of range", n, cchild[n]);
        uint32_t w[3] = {0, 0, 0};
        for (int i = 1; i <= lvl[n]; i++) {
            long long j;
            int b = base[(i > 1) ? 1 : 0];
            if (i > 1) { long long div = basediv[0]; for (int k = 0; k < i - 2; k++) div *= basediv[1]; j = ((cindex[n] / div) & (basediv[1] - 1)) - 1; }
            else j = (cindex[n] & (basediv[0] - 1)) - 1;
            if (j < 0) return fail(c, DRT_E_ARG, "drt_set_grid: cindex(%d) has no code at level %d", n, i);
            int v[3] = {(int)(j % b), (int)((j / b) % b), (int)((j / ((long long)b * b)) % b)};
            for (int k = 0; k < 3; k++) w[k] |= (uint32_t)v[k] << (4 * (i - 1));
        }
        c->cpack[3ull * n] = w[0]; c->cpack[3ull * n + 1] = w[1]; c->cpack[3ull * n + 2] = w[2];
    }
    int rc;
    if ((rc = dev_upload(c, &c->d_cchild, (const int*)cchild, (size_t)N))) return rc;
    if ((rc = dev_upload(c, &c->d_lvl, (const int*)lvl, (size_t)N))) return rc;
    if ((rc = dev_upload(c, &c->d_ccoord, ccoord, 3ull * N))) return rc;
    if ((rc = dev_upload(c, &c->d_csize, csize, (size_t)N))) return rc;
    if ((rc = dev_upload(c, &c->d_cpack, c->cpack.data(), 3ull * N))) return rc;
    // Face-neighbour table (DevProblem::nbr): for every leaf and face, carry the +-1 up the levels as increment does
    // (smooth_grid_routines.f90:751-769), walk from the root like ccindd_to_cc (:820-852) but no deeper than the leaf's level
    if (N > 0x07FFFFFF) return fail(c, DRT_E_ARG, "drt_set_grid: more than 2^27 cells");
    {
        std::vector<uint32_t> nbr(6ull * N, 0xFFFFFFFFu);
        for (int n = 1; n < N; n++) {
            if (cchild[n] != -1) continue;
            const int Lc = lvl[n];
            for (int f = 0; f < 6; f++) {
                const int ax = f >> 1, inc = (f & 1) ? 1 : -1;
                uint32_t w[3] = {c->cpack[3ull * n], c->cpack[3ull * n + 1], c->cpack[3ull * n + 2]};
                int k = inc;
                for (int j = Lc; j >= 1 && k != 0; j--) {
                    const int b = base[(j > 1) ? 1 : 0];
                    int v = (int)((w[ax] >> (4 * (j - 1))) & 15u) + k;
                    int ovr = k;
                    if (v >= b) v = 0; else if (v < 0) v = b - 1; else ovr = 0;
                    w[ax] = (w[ax] & ~(15u << (4 * (j - 1)))) | ((uint32_t)v << (4 * (j - 1)));
                    k = ovr;
                }
                if (k != 0) continue;  // out of the model
                int l = 1, found = -1, flv = 0;
                for (int i = 1; i <= Lc; i++) {
                    const int b = base[(i > 1) ? 1 : 0];
                    const int kk = ((int)((w[2] >> (4 * (i - 1))) & 15u) * b + (int)((w[1] >> (4 * (i - 1))) & 15u)) * b + (int)((w[0] >> (4 * (i - 1))) & 15u);
                    l += kk;
                    if (l < 1 || l >= N) break;
                    if (cchild[l] == -1 || i == Lc) { found = l; flv = i; break; }
                    l = cchild[l];
                }
                if (found < 0) return fail(c, DRT_E_ARG, "drt_set_grid: no face neighbour of cell %d on face %d (broken tree)", n, f);
                nbr[6ull * n + f] = (uint32_t)found | ((uint32_t)flv << 27) | (cchild[found] != -1 ? 0x80000000u : 0u);
            }
        }
        if ((rc = dev_upload(c, &c->d_nbr, nbr.data(), nbr.size()))) return rc;
    }
    c->have_grid = true; c->prepared = false; c->part2 = false;
    return DRT_OK;
}

extern "C" int drt_set_params(drt_ctx* c, const drt_params* p)
{
    if (!c || !p) return fail(c, DRT_E_ARG, "drt_set_params: bad argument");
    if (p->np_mpi < 1 || p->id_mpi < 0 || p->id_mpi >= p->np_mpi) return fail(c, DRT_E_ARG, "drt_set_params: bad np_mpi/id_mpi");
    if (p->bm_par <= 0 || p->bm_par_sca <= 0 || p->bm_par_max <= 0) return fail(c, DRT_E_ARG, "drt_set_params: bm_par* must be positive");
    if (p->kp_sca_max < 0 || p->kp_sca_max > 4) return fail(c, DRT_E_ARG, "drt_set_params: kp_sca_max outside [0,4] (check_input, io_routines.f90:2745)");
    if (c->comm && (p->np_mpi != c->nranks || p->id_mpi != c->rank)) return fail(c, DRT_E_ARG, "drt_set_params: np_mpi/id_mpi differ from the communicator's size/rank");
    c->par = *p; c->have_par = true; c->bm_par_cur = p->bm_par; c->prepared = false;
    return DRT_OK;
}

extern "C" int drt_set_wavelengths(drt_ctx* c, int32_t lnum, const double* ksca, const double* gsca)
{
    if (!c || lnum < 1 || lnum > DRT_MAXL || !ksca || !gsca) return fail(c, DRT_E_ARG, "drt_set_wavelengths: lnum must be within 1..%d", DRT_MAXL);
    c->L = lnum;
    c->LP = 1;
    while (c->LP < lnum) c->LP <<= 1;
    // several wavelengths: a lane owns a ray and loops over its wavelengths (DRT_RAY_MAJOR=0 keeps lane = (ray, wavelength))
    const char* rm = getenv("DRT_RAY_MAJOR");
    // DRT_RAY_MAJOR: unset = the ray-major kernel for every lnum; 0 = lane = (ray, wavelength) for every lnum; 2 = ray-major
    // for lnum >= 2 only (the selection of round 1: one wavelength went to rt_stage_kernel<1>)
    c->ray_major = !(rm && rm[0] == '0') && !(rm && rm[0] == '2' && lnum < 2);
    c->ksca.assign(ksca, ksca + lnum);
    c->gsca.assign(gsca, gsca + lnum);
    c->prepared = false;
    if (c->d_arena) { (void)hipFree(c->d_arena); c->d_arena = nullptr; }
    return DRT_OK;
}

extern "C" int drt_set_point_sources(drt_ctx* c, int32_t nsrc, const double* pos, const double* lum)
{
    if (!c || nsrc < 0 || (nsrc > 0 && (!pos || !lum))) return fail(c, DRT_E_ARG, "drt_set_point_sources: bad argument");
    if (c->L < 1) return fail(c, DRT_E_ARG, "drt_set_point_sources: call drt_set_wavelengths first");
    c->nsrc = nsrc;
    c->p_src_pos.assign(pos, pos + 3ull * nsrc);
    c->p_src_lum.assign(lum, lum + (size_t)nsrc * c->L);
    c->prepared = false;
    return DRT_OK;
}

extern "C" int drt_set_observers(drt_ctx* c, int32_t ndir, const double* dir_i_out)
{
    if (!c || ndir < 0 || (ndir > 0 && !dir_i_out)) return fail(c, DRT_E_ARG, "drt_set_observers: bad argument");
    c->ndir = ndir;
    c->dir_out.assign(dir_i_out, dir_i_out + 2ull * ndir);
    c->prepared = false;
    return DRT_OK;
}

// ---- helpers ------------------------------------------------------------------------------------------
// set_npix_arr (rt_routines.f90:5235-5315)
static void set_npix_arr(drt_ctx* c)
{
    const double pi = 2 * (double)asinf(1.0f);
    const int L = c->L;
    c->npix.assign(L, 0); c->npix_hp.assign(L, 0); c->kp.assign(L, 0); c->ik.assign(L, -1);
    const double g_th = 2 * (double)1e-3f;
    for (int i = 0; i < L; i++) {
        double g = fabs(c->gsca[i]);
        if (g < g_th) { c->npix_hp[i] = 1; c->kp[i] = -1; continue; }
        double pmax = (1 - g * g) / pow(1 + g * g - 2 * g, 1.5);
        double pmin = (1 - g * g) / pow(1 + g * g + 2 * g, 1.5);
        double cos_hm = (pow(2 * (1 - g * g) / (pmax + pmin), (double)(2.0f / 3.0f)) - (1 + g * g)) / (-2 * g);
        double fwhm = 2 * acos(cos_hm);
        double delta_min = fwhm / 5;
        int k_min = (int)(1 / (2 * (double)log10f(2.0f)) * log10(4 * pi / (12 * delta_min * delta_min)));
        if (k_min > c->par.kp_sca_max) k_min = c->par.kp_sca_max;
        c->kp[i] = k_min;
        c->npix_hp[i] = 12 * (1 << (2 * k_min));
    }
    for (int i = 0; i < L; i++) c->npix[i] = c->npix_hp[i] + c->ndir;  // tot_ndir_scaspe = tot_ndir (io_routines.f90:559)
    c->nstride.assign(L, 0);
    const char* ra = getenv("DRT_ROW_ALIGN");  // developer aid: 0 keeps the rows dense (A/B of the padded layout)
    for (int i = 0; i < L; i++) c->nstride[i] = (ra && ra[0] == '0') ? c->npix[i] : (c->npix[i] + 7) / 8 * 8;
    c->big_tables = false;
    for (int i = 0; i < L; i++) if (c->kp[i] >= 0 && c->npix[i] > 63) c->big_tables = true;
    c->nuniq = 0; c->kp_unique.clear(); c->npix_unique.clear();
    for (int k = 0; k <= c->par.kp_sca_max; k++) {
        bool inc = false;
        for (int i = 0; i < L; i++) if (c->npix_hp[i] == 12 * (1 << (2 * k))) inc = true;
        if (!inc) continue;
        for (int i = 0; i < L; i++) if (c->kp[i] == k) c->ik[i] = c->nuniq;
        c->kp_unique.push_back(k);
        c->npix_unique.push_back(12 * (1 << (2 * k)) + c->ndir);
        c->nuniq++;
    }
}

// host pix2ang_nest (healpix_routines.f90:30-136) for the scaspe direction tables (create_scaspe :2827-2848)
static void h_pix2ang_nest(int nside, int ipix, double* theta, double* phi)
{
    static const int jrll[12] = {2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4};
    static const int jpll[12] = {1, 3, 5, 7, 0, 2, 4, 6, 1, 3, 5, 7};
    const double halfpi = (double)asinf(1.0f);
    int npface = nside * nside, nl4 = 4 * nside, face_num = ipix / npface, ipf = ipix % npface;
    double fn = (double)nside, fact1 = 1.0 / (3.0 * fn * fn), fact2 = 2.0 / (3.0 * fn);
    int ix = 0, iy = 0;
    for (int b = 0; b < 15; b++) { ix |= ((ipf >> (2 * b)) & 1) << b; iy |= ((ipf >> (2 * b + 1)) & 1) << b; }
    int jrt = ix + iy, jpt = ix - iy, jr = jrll[face_num] * nside - jrt - 1, nr, kshift;
    double z;
    if (jr < nside) { nr = jr; z = 1.0 - nr * fact1 * nr; kshift = 0; }
    else if (jr <= 3 * nside) { nr = nside; z = (2 * nside - jr) * fact2; kshift = (jr - nside) & 1; }
    else { nr = nl4 - jr; z = -1.0 + nr * fact1 * nr; kshift = 0; }
    *theta = acos(z);
    int jp = (jpll[face_num] * nr + jpt + 1 + kshift) / 2;
    if (jp > nl4) jp -= nl4;
    if (jp < 1) jp += nl4;
    *phi = (jp - (kshift + 1) * 0.5) * (halfpi / nr);
}

// Direction tables (DevProblem::ring_tab / phi0_tab / phi1_tab): evaluated with the HOST libm so that
// pixel-centre sines and cosines are the very numbers the Fortran/CPU path computes
// (pix2ang_nest, healpix_routines.f90:104-129, then sin/cos in calc_cproj, rt_routines.f90:1945).
static int build_dir_tables(drt_ctx* c)
{
    if (c->tab_max_level > 0 && c->d_ring_tab[c->tab_max_level]) return DRT_OK;
    int kmax = 11;  // nside 2048: ~270 MB of tables; rays finer than that raise ERR_TABLE
    if (const char* e = getenv("DRT_TAB_NSIDE")) { int n = atoi(e); kmax = 2; while ((1 << (kmax + 1)) <= n && kmax < 13) kmax++; }
    const double halfpi = (double)asinf(1.0f);
    for (int k = 1; k <= kmax; k++) {
        const long nside = 1l << k, nl4 = 4 * nside;
        const double fn = (double)nside, fact1 = 1.0 / (3.0 * fn * fn), fact2 = 2.0 / (3.0 * fn);
        std::vector<double> ring(3ull * nl4, 0.0);
        for (long jr = 1; jr < nl4; jr++) {
            double z; long nr;
            if (jr < nside) { nr = jr; z = 1.0 - nr * fact1 * nr; }
            else if (jr <= 3 * nside) { nr = nside; z = (2 * nside - jr) * fact2; }
            else { nr = nl4 - jr; z = -1.0 + nr * fact1 * nr; }
            double th = acos(z);
            ring[3 * jr] = th; ring[3 * jr + 1] = sin(th); ring[3 * jr + 2] = cos(th);
        }
        std::vector<double> p0(3ull * 2 * nside * (nside + 1)), p1(3ull * nl4);
        // 3.4e7 sines and cosines for the levels up to nside 2048: a few seconds on one host core, so the rings are shared
        // among threads (every entry is its own libm call: the table does not depend on who computes it)
        auto fill_p0 = [&](long nr_lo, long nr_hi) {
            for (long nr = nr_lo; nr <= nr_hi; nr++)
                for (long jp = 1; jp <= 4 * nr; jp++) {
                    double ph = (jp - (0 + 1) * 0.5) * (halfpi / nr);
                    size_t o = 3ull * (2ull * nr * (nr - 1) + (jp - 1));
                    p0[o] = ph; p0[o + 1] = sin(ph); p0[o + 2] = cos(ph);
                }
        };
        const long nthr = (nside >= 256) ? std::max(1l, std::min(16l, (long)std::thread::hardware_concurrency())) : 1l;
        if (nthr <= 1) fill_p0(1, nside);
        else {
            // ring nr has 4 nr entries: equal shares of the entries = equal shares of nr^2
            std::vector<std::thread> th;
            long lo = 1;
            for (long t = 1; t <= nthr; t++) {
                const long hi = (t == nthr) ? nside : (long)(nside * sqrt((double)t / (double)nthr));
                if (hi >= lo) th.emplace_back(fill_p0, lo, hi);
                lo = std::max(lo, hi + 1);
            }
            for (auto& x : th) x.join();
        }
        for (long jp = 1; jp <= nl4; jp++) {
            double ph = (jp - (1 + 1) * 0.5) * (halfpi / nside);
            size_t o = 3ull * (jp - 1);
            p1[o] = ph; p1[o + 1] = sin(ph); p1[o + 2] = cos(ph);
        }
        int rc;
        if ((rc = dev_upload(c, &c->d_ring_tab[k], ring.data(), ring.size()))) return rc;
        if ((rc = dev_upload(c, &c->d_phi0_tab[k], p0.data(), p0.size()))) return rc;
        if ((rc = dev_upload(c, &c->d_phi1_tab[k], p1.data(), p1.size()))) return rc;
    }
    c->tab_max_level = kmax;
    return DRT_OK;
}

static int upload_problem(drt_ctx* c)
{
    DevProblem& P = c->hprob;
    memset(&P, 0, sizeof P);
    P.N = c->N; P.max_lvl = c->max_lvl; P.base0 = c->base[0]; P.base1 = c->base[1]; P.L = c->L; P.nsrc = c->nsrc; P.ndir = c->ndir;
    P.pabs_max = (int)(c->modelsize * 10);  // INTEGER pabs_max (rt_routines.f90:22, io_routines.f90:678)
    P.bm_par = c->bm_par_cur; P.bm_par_max = c->par.bm_par_max;
    for (int k = 0; k < 3; k++) P.wall_on[k] = c->par.wall_on[k];
    for (int k = 0; k < 6; k++) P.wall[k] = c->par.wall_coord[k];
    P.modelsize = c->modelsize; P.rad_lim_ms = c->par.rad_lim * c->modelsize; P.en_lim = c->en_lim; P.cs = c->par.cs;
    for (int i = 0; i <= c->max_lvl; i++) { P.csize_arr[i] = c->csize_arr[i]; P.carea_arr[i] = c->carea_arr[i]; P.cvol_arr[i] = c->cvol_arr[i]; }
    P.halfpi = (double)asinf(1.0f); P.pi = 2 * P.halfpi; P.twopi = 2 * P.pi; P.twothird = (double)(2.0f / 3.0f);
    P.glepsilon = (double)1.0e-7f; P.tau_thin = (double)1.0e-8f; P.len_eps = (double)1.0e-5f;
    P.outcube_f = (double)(1.0f + 1.0e-7f); P.merge_tol = (double)0.001f;
    P.cchild = c->d_cchild; P.lvl = c->d_lvl; P.ccoord = c->d_ccoord; P.csize = c->d_csize; P.cpack = c->d_cpack; P.nbr = c->d_nbr;
    P.dens = c->d_dens; P.u_fest = c->d_u_fest; P.u_final = c->d_u_final; P.lumcell = c->d_lumcell;
    P.p_src_pos = c->d_p_src_pos; P.p_src_lum = c->d_p_src_lum; P.src_cell = c->d_src_cell; P.dir_out = c->d_dir_out;
    P.lum_lost = c->d_lum_lost; P.i_obs = c->d_i_obs;
    for (int il = 0; il < c->L; il++) {
        P.ksca[il] = c->ksca[il]; P.g[il] = c->gsca[il];
        P.npix[il] = c->npix[il]; P.npix_hp[il] = c->npix_hp[il]; P.kp[il] = c->kp[il]; P.nstride[il] = c->nstride[il];
        P.scaspe[il] = il < (int)c->d_scaspe.size() ? c->d_scaspe[il] : nullptr;
        P.scaspe_prev[il] = il < (int)c->d_scaspe_prev.size() ? c->d_scaspe_prev[il] : nullptr;
        P.scaspe_tot[il] = il < (int)c->d_scaspe_tot.size() ? c->d_scaspe_tot[il] : nullptr;
        P.tab[il] = (c->ik[il] >= 0 && c->ik[il] < (int)c->d_tab.size()) ? c->d_tab[c->ik[il]] : nullptr;
        P.tabid[il] = P.tab[il] ? c->ik[il] : -1;
    }
    for (int u = 0; u < 4; u++) {
        P.utab[u] = (u < (int)c->d_tab.size()) ? c->d_tab[u] : nullptr;
        P.utab_np[u] = (u < (int)c->npix_unique.size()) ? c->npix_unique[u] : 0;
    }
    P.arena = c->d_arena;
    P.counters = c->d_counters; P.error = c->d_error; P.task_counter = c->d_task; P.src_cost = c->d_src_cost;
    P.dbg = c->d_dbg; P.dbg_cap = c->dbg_cap; P.dbg_flags = c->dbg_flags; if (const char* e = getenv("DRT_DBG_FLAGS")) P.dbg_flags |= atoi(e); P.iobs_healpix = (c->algorithm == DRT_RTA_I_OBS) ? 1 : 0;
    for (int k = 0; k < 14; k++) { P.ring_tab[k] = c->d_ring_tab[k]; P.phi0_tab[k] = c->d_phi0_tab[k]; P.phi1_tab[k] = c->d_phi1_tab[k]; }
    P.tab_max_level = c->tab_max_level;
    for (int k = 0; k < 14; k++) P.hgnorm[k] = c->d_hgnorm[k];
    P.hgnorm_max = c->hgnorm_max;
    P.dir_out_sc = c->d_dir_out_sc;
    HIPCHK(c, hipMemcpy(c->d_prob, &P, sizeof P, hipMemcpyHostToDevice));
    return DRT_OK;
}

// one wavelength: grids above ~5e4 cells run the three-waves-per-SIMD build of the kernel (rt_kernels.hpp)
static bool lp1_three_waves(const drt_ctx* c) { return c->LP == 1 && c->N > 50000 && !getenv("DRT_LP1_TWO_WAVES"); }

template <int LP> static int stage_blocks_per_cu()
{
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rt_stage_kernel<LP>, 256, 0) != hipSuccess || nb < 1) nb = 1;
    return nb;
}

static size_t wide_lds_bytes(const drt_ctx* c) { return (4ull * DRT_LDS_WORDS(c->L, c->LP) + (c->LP > 1 ? DRT_TAB_WORDS : 0)) * sizeof(double); }
template <int LP, int MODE> static int wide_blocks_per_cu_(drt_ctx* c)
{
    int nb = 0;
    const size_t lds = wide_lds_bytes(c);
    if (hipFuncSetAttribute((const void*)rt_stage_wide<LP, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) (void)hipGetLastError();
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rt_stage_wide<LP, MODE>, 256, lds) != hipSuccess || nb < 1) nb = 1;
    return nb;
}
// the smallest of the instantiations' answers (0: rows by atomics, 1: with the memory-table path, 2: deposit log): the arena is carved once for all
template <int LP> static int wide_blocks_per_cu(drt_ctx* c) { return std::min(wide_blocks_per_cu_<LP, 0>(c), std::min(wide_blocks_per_cu_<LP, 1>(c), wide_blocks_per_cu_<LP, 2>(c))); }

// The queue arena of a context that was not given one (drt_configure, or DRT_ARENA_GB in the environment): what keeps the depth of
// the ray queues - not their bytes - the same whatever the number of wavelengths a ray carries. A queue entry is 8 LP + 40 bytes (LP:
// the wavelengths per ray rounded up to a power of two); 16 GiB give 3072 waves ~18 000 rays per queue up to 4 wavelengths, and the
// same depth at 32 is 66 GiB - a quarter of the device's memory and 60 % of what is free at most. (Round 3: with a flat 16 GiB the first scattering call of the
// 32-wavelength bench grid was undone three to seven times, the direct light of the N-body grid took 28 s instead of 15.7 s.)
static size_t default_arena_bytes(const drt_ctx* c)
{
    if (const char* e = getenv("DRT_ARENA_GB")) { const double gb = atof(e); if (gb > 0) return (size_t)(gb * 1073741824.0); }
    const double base = 16.0 * 1073741824.0;
    double want = base * std::max(1.0, (8.0 * c->LP + 40.0) / 72.0);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0) {
        want = std::min(want, std::max(base, 0.25 * (double)total_b));
        want = std::min(want, std::max(4.0 * 1073741824.0, 0.6 * (double)free_b));   // several contexts on one device, other tenants: what is left decides
    } else (void)hipGetLastError();
    return ((size_t)want + 0xFFFFFull) & ~(size_t)0xFFFFF;
}

static int ensure_arena(drt_ctx* c)
{
    if (c->waves <= 0 && c->ray_major) {
        int nb = 1;
        switch (c->LP) {
        case 1: nb = wide_blocks_per_cu<1>(c); break;
        case 2: nb = wide_blocks_per_cu<2>(c); break;
        case 4: nb = wide_blocks_per_cu<4>(c); break;
        case 8: nb = wide_blocks_per_cu<8>(c); break;
        case 16: nb = wide_blocks_per_cu<16>(c); break;
        case 32: nb = wide_blocks_per_cu<32>(c); break;
        default: nb = wide_blocks_per_cu<64>(c); break;
        }
        c->waves = c->ncu * nb * 4;
    }
    if (c->waves <= 0) {
        // persistent waves: exactly as many as the chip keeps resident (more would only wait for a slot
        // while dividing the queue arena further)
        int nb = 1;
        switch (c->LP) {
        case 1:
            if (lp1_three_waves(c)) { if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rt_stage_kernel_1w3, 256, 0) != hipSuccess || nb < 1) nb = 1; }
            else nb = stage_blocks_per_cu<1>();
            break;
        case 2: nb = stage_blocks_per_cu<2>(); break;
        case 4: nb = stage_blocks_per_cu<4>(); break;
        case 8: nb = stage_blocks_per_cu<8>(); break;
        case 16: nb = stage_blocks_per_cu<16>(); break;
        case 32: nb = stage_blocks_per_cu<32>(); break;
        default: nb = stage_blocks_per_cu<64>(); break;
        }
        c->waves = c->ncu * nb * 4;
    }
    c->waves = (c->waves + 3) / 4 * 4;
    if (c->d_arena && c->arena_auto && c->arena_lp != c->LP) { (void)hipFree(c->d_arena); c->d_arena = nullptr; }   // sized for other wavelengths
    if (c->d_arena) return DRT_OK;
    if (c->arena_auto) { c->arena_bytes = default_arena_bytes(c); c->arena_lp = c->LP; }
    HIPCHK(c, hipMalloc((void**)&c->d_arena, c->arena_bytes));
    return DRT_OK;
}

// Split the arena among the `waves` wavefronts of one launch: the deepest queues (`cap` rays, a multiple of 64) whose
// slices (arena_layout, rt_kernels.hpp) fit. Pure arithmetic, exported so that the layout can be checked without a GPU.
#define DRT_MIN_CAP 64
extern "C" int drt_arena_carve(size_t arena_bytes, int32_t waves, int32_t lp, int32_t* cap_out, int32_t* hcap_out, uint64_t* stride_out,
                               uint64_t offsets[10])
{
    if (waves < 1 || lp < 1 || lp > DRT_MAXL || !cap_out || !hcap_out || !stride_out) return DRT_E_ARG;
    const unsigned long long per_ray = 4ull * (28 + 8ull * lp) + 4ull * 20 + 8 + 16;  // lists + hash (2 entries per ray) + codes + status
    const unsigned long long per_wave = arena_bytes / (unsigned long long)waves;
    if (per_wave < 1024ull * lp + 272 + DRT_MIN_CAP * per_ray) return DRT_E_CAPACITY;
    unsigned long long cap = (per_wave - 1024ull * lp - 272) / per_ray;
    if (cap > (1ull << 22)) cap = 1ull << 22;
    cap = cap / 64 * 64;
    for (;; cap -= 64) {
        if (cap < DRT_MIN_CAP) return DRT_E_CAPACITY;
        int hcap = 64;
        while (hcap < 2 * (int)cap) hcap <<= 1;
        const ArenaLayout L = arena_layout((int)cap, hcap, lp);
        const unsigned long long stride = (L.end + 255) / 256 * 256;
        if (stride * (unsigned long long)waves <= arena_bytes) {
            *cap_out = (int32_t)cap; *hcap_out = hcap; *stride_out = stride;
            if (offsets) {
                for (int k = 0; k < 4; k++) offsets[k] = L.list[k];
                offsets[4] = L.hkey; offsets[5] = L.hmem; offsets[6] = L.ostat; offsets[7] = L.thg; offsets[8] = L.inten; offsets[9] = L.end;
            }
            return DRT_OK;
        }
    }
}

// test aid: DRT_TEST_MAX_CAP shortens the queues (>= 16 rays) so that small cases exercise the overflow paths; the
// slices keep their stride, only less of each is used (>= 16 rays, >= 4 for one-sector tasks)
static void test_max_cap(int sectors_per_task, int32_t& cap, int32_t& hcap)
{
    if (const char* e = getenv("DRT_TEST_MAX_CAP")) {
        const int lim = std::max(sectors_per_task > 1 ? 16 : 4, atoi(e));  // a task starts with sectors_per_task rays in its queue
        if (cap > lim) { cap = lim; hcap = 64; while (hcap < 2 * cap) hcap <<= 1; }
    }
}

// rays per queue when `waves` wavefronts share the arena (0: too small for that many)
static int queue_cap(const drt_ctx* c, int waves, int sectors_per_task)
{
    int32_t cap = 0, hcap = 0;
    uint64_t stride = 0;
    if (drt_arena_carve(c->arena_bytes, waves, c->LP, &cap, &hcap, &stride, nullptr) != DRT_OK) return 0;
    test_max_cap(sectors_per_task, cap, hcap);
    return cap;
}

static int carve_arena(drt_ctx* c, int waves, StageArgs& a)
{
    int32_t cap = 0, hcap = 0;
    uint64_t stride = 0;
    if (drt_arena_carve(c->arena_bytes, waves, c->LP, &cap, &hcap, &stride, nullptr) != DRT_OK)
        return fail(c, DRT_E_CAPACITY, "arena of %zu bytes too small for %d waves (fewer than %d rays per queue)", c->arena_bytes, waves, DRT_MIN_CAP);
    test_max_cap(a.sectors_per_task, cap, hcap);
    a.cap = cap; a.hcap = hcap; a.wave_stride = stride;
    c->cap = a.cap; c->hcap = a.hcap; c->wave_stride = a.wave_stride;
    return DRT_OK;
}

static const char* trace_error_text(int e)
{
    if (e & ERR_CAPACITY) return "a ray queue outgrew its arena (raise arena_bytes or lower waves with drt_configure)";
    if (e & ERR_NEG_LENGTH) return "negative ray-cell intersection length (rt_routines.f90:1714)";
    if (e & ERR_CELL_NOT_FOUND) return "next cell not found (rt_routines.f90:1851)";
    if (e & ERR_RELOCATE) return "host cell of a re-launched ray not found (rt_routines.f90:2031-2063)";
    if (e & ERR_NSIDE) return "nside out of range (healpix_routines.f90:54)";
    if (e & ERR_STEPS) return "step limit reached";
    if (e & ERR_MERGE) return "inconsistent sibling table while merging low rays";
    if (e & ERR_TABLE) return "ray finer than the HEALPix direction tables (set DRT_TAB_NSIDE to a larger power of two)";
    if (e & ERR_LOG) return "the deposit log is full (raise DRT_LOG_GB)";
    return "unknown device error";
}

// ---- rt_prepare -------------------------------------------------------------------------------------------
static int prepare_start(drt_ctx* c)
{
    if (!c->have_grid || !c->have_par || c->L < 1) return fail(c, DRT_E_ARG, "rt_prepare(start): grid, params and wavelengths must be set first");
    if (!c->d_dens || c->dens_stars_h.size() != (size_t)c->N * c->L) return fail(c, DRT_E_ARG, "rt_prepare(start): dens_arr and dens_stars_arr must be set first (drt_set_array)");
    const int N = c->N, L = c->L;
    const bool timing = getenv("DRT_TIMING") != nullptr;
    auto tnow = [] { return std::chrono::steady_clock::now(); };
    auto tms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t_0 = tnow();
    set_npix_arr(c);
    // set_en_lim (rt_routines.f90:3137-3158)
    int leaves = 0;
    for (int i = 0; i < N; i++) leaves += (c->cchild[i] == -1);
    c->en_lim = c->par.accuracy / ((leaves + c->nsrc) * 0.25);
    // prepare_p_src + fix_ccoord_p_src (rt_routines.f90:535-601)
    const double tol_p = (double)1.0e-6f;
    c->src_cell.assign(N, -1);
    c->cell_src.assign(c->nsrc > 0 ? c->nsrc : 1, 0);
    for (int i = 0; i < c->nsrc; i++) {
        bool found = false;
        for (int j = 0; j < N && !found; j++) {
            if (c->cchild[j] != -1) continue;
            double rel[3], cellsize = c->csize[j];
            int in = 0;
            for (int k = 0; k < 3; k++) { rel[k] = c->p_src_pos[3ull * i + k] - c->ccoord[3ull * j + k]; if (fabs(rel[k]) < (1. + tol_p) * cellsize / 2.) in++; }
            if (in == 3) {
                c->src_cell[j] = i; c->cell_src[i] = j;
                for (int k = 0; k < 3; k++) if (fabs(rel[k]) >= cellsize / 2.) {
                    if (rel[k] > 0) c->p_src_pos[3ull * i + k] -= tol_p * cellsize; else c->p_src_pos[3ull * i + k] += tol_p * cellsize;
                }
                found = true;
            }
        }
        if (!found) return fail(c, DRT_E_ARG, "host cell of point source %d not found (prepare_p_src, rt_routines.f90:567)", i);
    }
    // calc_total_luminosity (rt_routines.f90:605-660)
    c->lumcell_h.assign((size_t)N * L, 0.0);
    for (int i = 0; i < N; i++) {
        if (c->cchild[i] != -1) continue;
        double v = c->csize[i] * c->csize[i] * c->csize[i];
        for (int il = 0; il < L; il++) c->lumcell_h[(size_t)i * L + il] = c->dens_stars_h[(size_t)i * L + il] * v;
    }
    c->tot_rad_en.assign(L, 0.0);
    int rc;
    if ((rc = dev_upload(c, &c->d_lumcell, c->lumcell_h.data(), (size_t)N * L))) return rc;
    if ((rc = dev_upload(c, &c->d_src_cell, (const int*)c->src_cell.data(), (size_t)N))) return rc;
    if ((rc = dev_upload(c, &c->d_p_src_pos, c->p_src_pos.data(), 3ull * c->nsrc))) return rc;
    if ((rc = dev_upload(c, &c->d_p_src_lum, c->p_src_lum.data(), (size_t)c->nsrc * L))) return rc;
    if ((rc = dev_upload(c, &c->d_dir_out, c->dir_out.data(), 2ull * c->ndir))) return rc;
    {
        std::vector<double> sc(4ull * (c->ndir > 0 ? c->ndir : 1), 0.0);
        for (int j = 0; j < c->ndir; j++) {
            double th = c->dir_out[j], ph = c->dir_out[c->ndir + j];
            sc[4ull * j] = sin(th); sc[4ull * j + 1] = cos(th); sc[4ull * j + 2] = sin(ph); sc[4ull * j + 3] = cos(ph);
        }
        if ((rc = dev_upload(c, &c->d_dir_out_sc, sc.data(), sc.size()))) return rc;
    }
    const auto t_1 = tnow();
    if ((rc = build_dir_tables(c))) return rc;
    const auto t_2 = tnow();
    if ((rc = dev_zero(c, &c->d_u_fest, (size_t)N * L))) return rc;
    if ((rc = dev_zero(c, &c->d_u_final, (size_t)N * L))) return rc;
    if ((rc = dev_zero(c, &c->d_lum_lost, (size_t)L))) return rc;
    if ((rc = dev_zero(c, &c->d_lum_lost_prev, (size_t)L))) return rc;
    if ((rc = dev_zero(c, &c->d_src_cost, 2 * (size_t)N))) return rc;   // costs, then queue lengths (DevProblem::src_cost)
    c->src_q_h.assign((size_t)N, 0.0);
    free_fields(c);
    c->iterations = 0; c->cnflag = 0; c->bm_par_cur = c->par.bm_par; c->part2 = false;
    c->prepared = true;
    if ((rc = ensure_arena(c))) return rc;
    rc = upload_problem(c);
    if (timing) fprintf(stderr, "[dartray_rt] rt_prepare(start): sources and uploads %.1f ms, direction tables %.1f ms, fields and arena %.1f ms\n", tms(t_0, t_1), tms(t_1, t_2), tms(t_2, tnow()));
    return rc;
}

// create_scaspe + create_i_obs (rt_routines.f90:2784-2852, 2970-3009)
static int prepare_part2(drt_ctx* c)
{
    if (!c->prepared) return fail(c, DRT_E_ARG, "rt_prepare(prep_part2) before rt_prepare(start)");
    const int N = c->N, L = c->L;
    free_fields(c);
    c->d_scaspe.assign(L, nullptr); c->d_scaspe_prev.assign(L, nullptr); c->d_scaspe_tot.assign(L, nullptr);
    int rc;
    for (int il = 0; il < L; il++) if ((rc = dev_zero(c, &c->d_scaspe[il], (size_t)N * c->nstride[il]))) return rc;
    c->d_tab.assign(c->nuniq, nullptr);
    for (int u = 0; u < c->nuniq; u++) {
        int n = c->npix_unique[u], nhp = n - c->ndir, nside_sca = 1 << c->kp_unique[u];
        // unit vectors of the scaspe slots: HEALPix pixel centres, then the observer directions
        // (create_scaspe, rt_routines.f90:2827-2848); calc_ads_arr's cosine (:2294) is their dot product
        // with the ray direction
        std::vector<double> t(3ull * n);
        for (int j = 0; j < n; j++) {
            double th, ph;
            if (j < nhp) h_pix2ang_nest(nside_sca, j, &th, &ph);
            else { th = c->dir_out[j - nhp]; ph = c->dir_out[c->ndir + (j - nhp)]; }
            t[j] = sin(th) * cos(ph); t[n + j] = sin(th) * sin(ph); t[2ull * n + j] = cos(th);
        }
        if ((rc = dev_upload(c, &c->d_tab[u], t.data(), t.size()))) return rc;
    }
    if ((rc = dev_zero(c, &c->d_i_obs, (size_t)(c->ndir > 0 ? c->ndir : 1) * (N + c->nsrc) * L))) return rc;
    c->part2 = true;
    // HG normalisation per ray direction (DevProblem::hgnorm): levels nside 4 .. 256 by default (268 MB at 32 wavelengths;
    // DRT_HGNORM_NSIDE changes the finest tabulated nside, 0 switches the table off)
    for (int k = 0; k < 14; k++) if (c->d_hgnorm[k]) { (void)hipFree(c->d_hgnorm[k]); c->d_hgnorm[k] = nullptr; }
    c->hgnorm_max = 0;
    if ((rc = upload_problem(c))) return rc;
    int kmax = 8;
    if (const char* e = getenv("DRT_HGNORM_NSIDE")) { const int n = atoi(e); kmax = 0; while (kmax < 11 && (1 << (kmax + 1)) <= n) kmax++; if (n < 4) kmax = 0; }
    kmax = std::min(kmax, c->tab_max_level);
    bool any_hg = false;
    for (int il = 0; il < L; il++) if (c->ik[il] >= 0) any_hg = true;
    if (any_hg && kmax >= 2) {
        for (int k = 2; k <= kmax; k++) {
            const long long n = (12ll << (2 * k)) * L;
            HIPCHK(c, hipMalloc((void**)&c->d_hgnorm[k], (size_t)n * sizeof(double)));
            hipLaunchKernelGGL(k_hgnorm, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, c->d_prob, k, c->d_hgnorm[k]);
        }
        HIPCHK(c, hipDeviceSynchronize());
        c->hgnorm_max = kmax;
    }
    return upload_problem(c);
}

// u_final sums that rode in the padding slot of the scaspe rows (rt_wide.hpp, flush_lambda) go where they belong
__global__ void k_fold_pad(double* __restrict__ row, double* __restrict__ u_final, int N, int L, int il, int np, int ns)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const double v = row[(size_t)i * ns + np];
    if (v != 0.0) { u_final[(size_t)i * L + il] += v; row[(size_t)i * ns + np] = 0.0; }
}

static int lumcell_from(drt_ctx* c, const std::vector<double*>& arr, int all_pix, std::vector<double>& totals)
{
    const int N = c->N, L = c->L;
    HIPCHK(c, hipMemset(c->d_tot, 0, DRT_MAXL * sizeof(double)));
    for (int il = 0; il < L; il++)
        hipLaunchKernelGGL(k_lumcell_sca, dim3((N + 255) / 256), dim3(256), 0, 0, arr[il], c->d_cchild, c->d_lumcell, c->d_tot + il, N, L, il,
                           c->npix[il], c->npix_hp[il], all_pix, c->nstride[il]);
    totals.assign(L, 0.0);
    HIPCHK(c, hipMemcpy(totals.data(), c->d_tot, L * sizeof(double), hipMemcpyDeviceToHost));
    c->lumcell_h.resize((size_t)N * L);
    HIPCHK(c, hipMemcpy(c->lumcell_h.data(), c->d_lumcell, (size_t)N * L * sizeof(double), hipMemcpyDeviceToHost));
    return DRT_OK;
}

// rt_prepare, case rtt_scatt (dartray_hub.f90:432-457) + calc_total_luminosity_sca (rt_routines.f90:663-758)
static int prepare_scatt(drt_ctx* c)
{
    if (!c->part2) return fail(c, DRT_E_ARG, "rt_prepare(scatt) before rt_prepare(prep_part2)");
    const int N = c->N, L = c->L;
    const size_t NL = (size_t)N * L;
    HIPCHK(c, hipMemcpy(c->d_u_fest, c->d_u_final, NL * sizeof(double), hipMemcpyDeviceToDevice));
    HIPCHK(c, hipMemcpy(c->d_lum_lost_prev, c->d_lum_lost, L * sizeof(double), hipMemcpyDeviceToDevice));
    c->iterations++;
    int rc;
    if (c->iterations == 1) {
        for (int il = 0; il < L; il++) {
            size_t n = (size_t)N * c->nstride[il];
            if ((rc = dev_zero(c, &c->d_scaspe_tot[il], n))) return rc;
            HIPCHK(c, hipMemcpy(c->d_scaspe_tot[il], c->d_scaspe[il], n * sizeof(double), hipMemcpyDeviceToDevice));  // create_scaspe_tot
            if ((rc = dev_zero(c, &c->d_scaspe_prev[il], n))) return rc;                                             // create_scaspe_prev
        }
        c->bm_par_cur = c->par.bm_par_sca;
    }
    for (int il = 0; il < L; il++)
        HIPCHK(c, hipMemcpy(c->d_scaspe_prev[il], c->d_scaspe[il], (size_t)N * c->nstride[il] * sizeof(double), hipMemcpyDeviceToDevice));
    std::vector<double> tot;
    if ((rc = lumcell_from(c, c->d_scaspe, 0, tot))) return rc;
    if (c->iterations == 1) c->tot_rad_en = tot;  // parents hold zeros, so the leaf sum equals sum over all cells (:682)
    double sum_tot = 0;
    for (int il = 0; il < L; il++) sum_tot += c->tot_rad_en[il];
    if (sum_tot > 0) {
        bool all = true;
        for (int il = 0; il < L; il++) if (!(tot[il] / c->tot_rad_en[il] < c->par.conv_en_lim)) all = false;
        c->cnflag = all ? 1 : 0;
    } else c->cnflag = 1;
    if (c->par.max_sca_iterations >= 0 && c->iterations > c->par.max_sca_iterations) c->cnflag = 1;
    // sequential scattering bookkeeping (:747-756)
    for (int il = 0; il < L; il++) {
        size_t n = (size_t)N * c->nstride[il];
        if (c->iterations > 1) hipLaunchKernelGGL(k_axpy, dim3(1024), dim3(256), 0, 0, c->d_scaspe_tot[il], c->d_scaspe[il], (unsigned long long)n);
        HIPCHK(c, hipMemsetAsync(c->d_scaspe[il], 0, n * sizeof(double), 0));
    }
    HIPCHK(c, hipDeviceSynchronize());
    return upload_problem(c);
}

extern "C" int drt_rt_prepare(drt_ctx* c, int32_t rt_type)
{
    if (!c) return DRT_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    switch (rt_type) {
    case DRT_RTT_START: return prepare_start(c);
    case DRT_RTT_PREP_PART2: return prepare_part2(c);
    case DRT_RTT_SCATT: return prepare_scatt(c);
    case DRT_RTT_I_OBS: {  // calc_total_luminosity_sca, case rtt_i_obs (rt_routines.f90:760-771)
        if (c->d_scaspe_tot.empty() || !c->d_scaspe_tot[0]) return fail(c, DRT_E_ARG, "rt_prepare(i_obs): no scaspe_tot (run the scattering stage first)");
        std::vector<double> tot;
        return lumcell_from(c, c->d_scaspe_tot, 1, tot);
    }
    default: return fail(c, DRT_E_ARG, "drt_rt_prepare: rt_type %d not handled", rt_type);
    }
}

// Longest-processing-time deal of chunks of consecutive sources among np ranks, by expected cost: the chunks (at most `chunk`
// sources) keep neighbours together, a source that costs more than 1/64 of a rank's share is a chunk of its own; heaviest chunk
// first to the least loaded rank (ties: the earlier chunk, the lower rank). A pure function of its arguments: every rank computes
// the same deal from the same costs. owner[k] = rank of source k. (drt_deal_by_cost exposes it for the CPU tests.)
static void deal_by_cost(const double* cost, size_t n, int np, size_t chunk, std::vector<int>& owner)
{
    owner.assign(n, 0);
    if (np <= 1 || n == 0) return;
    if (chunk < 1) chunk = 1;
    double csum = 0;
    for (size_t k = 0; k < n; k++) csum += cost[k];
    const double big = csum / np / 64.0;
    struct Chunk { size_t lo, hi; double cost; };
    std::vector<Chunk> ch;
    for (size_t k = 0; k < n;) {
        Chunk q{k, k, 0.0};
        while (q.hi < n && q.hi - q.lo < chunk) {
            const double v = cost[q.hi];
            if (v > big) { if (q.hi == q.lo) { q.cost = v; q.hi++; } break; }
            q.cost += v; q.hi++;
        }
        ch.push_back(q);
        k = q.hi;
    }
    std::vector<size_t> order(ch.size());
    for (size_t k = 0; k < order.size(); k++) order[k] = k;
    std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return ch[x].cost > ch[y].cost; });
    std::vector<double> load((size_t)np, 0.0);
    for (size_t k : order) {
        int r = 0;
        for (int q = 1; q < np; q++) if (load[(size_t)q] < load[(size_t)r]) r = q;
        load[(size_t)r] += ch[k].cost;
        for (size_t m = ch[k].lo; m < ch[k].hi; m++) owner[m] = r;
    }
}

extern "C" int drt_deal_by_cost(const double* cost, int64_t n, int32_t np, int32_t chunk, int32_t* owner)
{
    if (!cost || !owner || n < 0 || np < 1) return DRT_E_ARG;
    std::vector<int> o;
    deal_by_cost(cost, (size_t)n, np, (size_t)std::max(1, (int)chunk), o);
    for (int64_t k = 0; k < n; k++) owner[k] = o[(size_t)k];
    return DRT_OK;
}

// ---- queue lengths of the sources nothing is known of, from the ones an undone launch measured (pure; drt_fit_queue_prior) -------
// A source's longest queue grows with its luminosity: log q48 against log luminosity is a line of slope 0.5-1.1 with correlation
// 0.62-0.95 in the cell-source stages of the BASELINE configurations, 99 % of the sources within a factor 2-4 of it
// (tools/queue_vs_cost.py). Least squares in the logarithms over the n pairs with q > 0 and lum > 0; slope kept within [0, 1.5].
// False (nothing to go by) with fewer than 64 pairs or luminosities that are all alike.
static bool fit_queue_prior(const double* lum, const double* q, size_t n, double& a, double& b)
{
    double sx = 0, sy = 0, sxx = 0, sxy = 0; size_t m = 0;
    for (size_t k = 0; k < n; k++) if (q[k] > 0 && lum[k] > 0) { const double x = log(lum[k]), y = log(q[k]); sx += x; sy += y; sxx += x * x; sxy += x * y; m++; }
    if (m < 64) return false;
    const double vx = sxx / m - (sx / m) * (sx / m);
    if (!(vx > 1e-6)) return false;
    b = (sxy / m - (sx / m) * (sy / m)) / vx;
    b = std::min(1.5, std::max(0.0, b));
    a = sy / m - b * sx / m;
    return true;
}
extern "C" int drt_fit_queue_prior(const double* lum, const double* q, int64_t n, double* a, double* b)
{
    if (!lum || !q || !a || !b || n < 0) return DRT_E_ARG;
    return fit_queue_prior(lum, q, (size_t)n, *a, *b) ? DRT_OK : DRT_E_CAPACITY;
}

// ---- task planning of a cell-source launch (pure; drt_plan_tasks exposes it for the CPU tests) -----------------------------
// A task is one wave's serial work: no task may be a large part of its launch. From the expected cost of every source (measured
// wave time scaled with the luminosity, or the prior sqrt(luminosity)), with limit = (sum of the costs / waves) / hd / nbatch:
//  * deep sources  - one sector (1/48 of the source) would exceed the limit - are run as sub-tree tasks, before the others;
//  * heavy sources - a task of G sectors would exceed it - get fewer sectors per task (a power-of-two fraction of G, down to one),
//    their tasks listed explicitly, heaviest source first;
//  * light sources take G sectors per task.
// hd: base 8 (DRT_HEAVY_DIV); x 4 where nothing is measured yet (the prior's ratios are good, but it knows nothing of queue
// lengths); and never coarser than the queues allow: the stage's last launch had its limit at f_last of a wave's share with G_last
// sectors and filled q_fill of the queue capacity, so this one may go up to f_last * 0.8 / q_fill * G_last / G.
// Queue lengths, where the sources' own are known (q48[i]: the longest queue a task of all 48 sectors of source i would hold, from
// its tasks in the stage's last launch; queues grow with the sectors of a task, whatever its cost): a source takes the sectors per
// task that keep its queue below 0.8 of the capacity `cap` - a heavy source by that count too - and is deep where one sector
// does not. This replaces the rule on f_last, which knows one queue of the whole launch (TRUST-I tau = 1, 21 801 cells, 32 wavelengths,
// 16 GiB arena: the second scattering call, now measured, held the limit that rule allowed, overflowed at 8 and at 4 sectors and ran
// at 2).
// No source is deep where the costs are measured and a wave's share of the launch is below deep_min cycles (the sub-tree scheme
// is two more launches).
struct PlanIn {
    const int* list; size_t n; const double* cost;   // cost[list[k]]
    int waves, G, nbatch;
    double heavy_div, deep_min;
    bool measured, no_deep;
    double f_last, q_fill; int G_last;
    const double* q48 = nullptr; double cap = 0;   // q48[list[k]], 0 = not known
    double stage_total = 0;   // > 0: the list is one part of a stage with this sum of costs, and the limit is the stage's, not the part's
};
struct PlanOut { std::vector<int> light, deep; std::vector<int2> heavy; double f_used = 0, limit = 0, total = 0; int nbatch_sub = 0; size_t n_heavy_src = 0, n_queue_cut = 0; };
static void plan_tasks(const PlanIn& in, PlanOut& o)
{
    o.light.clear(); o.heavy.clear(); o.deep.clear(); o.f_used = 0; o.limit = 0; o.nbatch_sub = 0; o.n_heavy_src = 0; o.n_queue_cut = 0;
    double total = 0;
    for (size_t k = 0; k < in.n; k++) total += in.cost[(size_t)in.list[k]];
    o.total = total;
    if (!(in.heavy_div > 0) || !(total > 0)) { o.light.assign(in.list, in.list + in.n); return; }
    double hd = in.heavy_div * (in.measured ? 1.0 : 4.0);
    bool own_queues = false;
    if (in.q48 && in.cap > 0) for (size_t k = 0; k < in.n && !own_queues; k++) own_queues = in.q48[(size_t)in.list[k]] > 0;
    if (!own_queues && in.f_last > 0 && in.q_fill > 0 && in.G_last > 0) {
        const double f_max = in.f_last * 0.8 / in.q_fill * (double)in.G_last / (double)in.G;
        if (1.0 / hd > f_max) hd = 1.0 / f_max;
    }
    o.f_used = 1.0 / hd;
    const double limit = (in.stage_total > total ? in.stage_total : total) / (double)in.waves / hd / (double)std::max(1, in.nbatch);
    o.limit = limit;
    const bool no_deep = in.no_deep || (in.measured && total / (double)in.waves / (double)std::max(1, in.nbatch) < in.deep_min);
    std::vector<std::pair<double, int>> hv;
    double dsum = 0;
    // sectors per task the queue of source i allows (0: not even one)
    auto g_queue = [&](int i) { int g = in.G; if (own_queues) { const double q = in.q48[(size_t)i]; while (g > 0 && q * g / 48.0 > 0.8 * in.cap) g >>= 1; } return g; };
    for (size_t k = 0; k < in.n; k++) {
        const int i = in.list[k];
        const double v = in.cost[(size_t)i];
        const int gq = g_queue(i);
        if (gq < in.G) o.n_queue_cut++;
        if ((v / 48.0 > limit && !no_deep) || (gq == 0 && !in.no_deep)) { o.deep.push_back(i); dsum += v; }
        else if (v * in.G / 48.0 > limit || gq < in.G) hv.push_back({v, i});
        else o.light.push_back(i);
    }
    // the deep sources' share of the records decides how many batches their sub-tree launches need (deposit log)
    o.nbatch_sub = std::max(1, std::min(std::max(1, in.nbatch), (int)ceil(2.0 * std::max(1, in.nbatch) * dsum / total)));
    std::stable_sort(hv.begin(), hv.end(), [](const std::pair<double, int>& x, const std::pair<double, int>& y) { return x.first > y.first; });
    o.n_heavy_src = hv.size();
    for (auto& h : hv) {
        int Gi = std::max(1, g_queue(h.second));
        while (Gi > 1 && h.first * Gi / 48.0 > limit) Gi >>= 1;
        for (int s0 = 0; s0 < 48; s0 += Gi) o.heavy.push_back(make_int2(h.second, s0 | (Gi << 8)));
    }
}

// cls[k] of source k: 0 light, -1 deep, else the sectors per task of a heavy source. Returns the number of explicit heavy tasks.
extern "C" int drt_plan_tasks(const double* cost, int64_t n, int32_t waves, int32_t G, int32_t nbatch, double heavy_div, double deep_min,
                              int32_t measured, double f_last, double q_fill, int32_t G_last, const double* q48, double cap, int32_t* cls, double* limit_out)
{
    if (!cost || !cls || n < 0 || waves < 1 || (G != 1 && G != 2 && G != 4 && G != 8 && G != 16)) return DRT_E_ARG;
    std::vector<int> list((size_t)n);
    for (int64_t k = 0; k < n; k++) list[(size_t)k] = (int)k;
    PlanIn in{list.data(), (size_t)n, cost, waves, G, std::max(1, (int)nbatch), heavy_div, deep_min, measured != 0, false, f_last, q_fill, G_last, q48, cap};
    PlanOut o;
    plan_tasks(in, o);
    for (int64_t k = 0; k < n; k++) cls[k] = -2;
    for (int i : o.light) cls[i] = 0;
    for (int i : o.deep) cls[i] = -1;
    for (const int2& h : o.heavy) cls[h.x] = h.y >> 8;
    if (limit_out) *limit_out = o.limit;
    return (int)o.heavy.size();
}

// ---- rt_loop ---------------------------------------------------------------------------------------------
// The sources of one rt_loop call on this rank. Cell sources are shared among the ranks in CHUNKS of consecutive
// emitting cells (DRT_PARTITION_CHUNK cells per chunk, default 64), dealt by expected cost (deal_by_cost; DRT_PARTITION=cyclic
// deals them round the ranks instead). Sums over sources commute, so every deal gives the same fields after rt_mpi; what
// differs is speed (measured on the bench workload with 8 ranks, tools/partition_ab.py):
//  * the reference's deal, every np-th cell (i = im*np_mpi + id_mpi, rt_routines.f90:1003-1005; DRT_PARTITION=mod),
//    balances well but a rank's sources are no longer neighbours: they stop sharing tree, density and scaspe lines in
//    the caches and the SUM of the ranks' times grows by 40 % at 32 wavelengths and 5.4x at one;
//  * one contiguous block per rank (DRT_PARTITION=block) keeps the locality but the cost of a source varies with its
//    place in the model - the slowest rank had 3.9x the mean work;
//  * chunks keep neighbours together (a GPU has ~170 sources in flight at once) and sample the whole model.
// Point sources are not shared here: every rank takes all of them and the ranks split their (source, sector) tasks
// instead (StageArgs::task_first/task_stride). rt_loop_iobs runs on the main rank only (rt_routines.f90:1483, :1498).
static int build_src_list(drt_ctx* c, int rt_type, std::vector<int>& list, bool launching = false)
{
    const int N = c->N, L = c->L, np = c->par.np_mpi, id = c->par.id_mpi;
    list.clear();
    const bool psrc = (rt_type == DRT_RTT_PRECALC_SRC || rt_type == DRT_RTT_DIR_SRC || rt_type == DRT_RTT_I_OBS_DIR_SRC);
    const bool iobs = (rt_type == DRT_RTT_I_OBS_DIR_CELL || rt_type == DRT_RTT_I_OBS_DIR_SRC || rt_type == DRT_RTT_I_OBS);
    const char* pe = getenv("DRT_PARTITION");
    const bool by_mod = pe && !strcmp(pe, "mod"), by_block = pe && !strcmp(pe, "block");
    size_t chunk = 64;
    if (const char* ce = getenv("DRT_PARTITION_CHUNK")) chunk = (size_t)std::max(1, atoi(ce));
    for (int i = 0; i < N; i++) {
        if (c->cchild[i] != -1) continue;
        double s = 0;
        if (psrc) { if (c->src_cell[i] != -1) for (int il = 0; il < L; il++) s += c->p_src_lum[(size_t)c->src_cell[i] * L + il]; }
        else for (int il = 0; il < L; il++) s += c->lumcell_h[(size_t)i * L + il];
        if (s == 0) continue;  // rt_routines.f90:1016
        if (!iobs && !psrc && by_mod && (i % np) != id) continue;
        list.push_back(i);
    }
    // Expected cost of every source of this launch. Measured (same kind of stage, cost known): the measured wave time times
    // the ratio of the source's luminosity now and then - a source emits fewer rays as its luminosity falls from one
    // scattering iteration to the next (bench grid: the time of a source falls to 8-34 % from iteration 1 to 2, exponent 1
    // predicts the ranks' shares to 4-19 %; unchanged luminosity, as in a repeated step: the measurement itself).
    // Otherwise a prior, sqrt(luminosity), scaled to the measured ones if there are any: log cost against log luminosity
    // is a line of slope 0.5-0.9 with correlation 0.91-0.98 on the bench grid. Only ratios of these numbers are used.
    if (!iobs && !psrc && c->d_src_cost) {
        std::vector<double> meas(2 * (size_t)N);
        HIPCHK(c, hipMemcpy(meas.data(), c->d_src_cost, 2 * (size_t)N * sizeof(double), hipMemcpyDeviceToHost));
        c->src_pred_h.assign((size_t)N, 0.0);
        if (c->cost_lum_h.size() != (size_t)N) c->cost_lum_h.assign((size_t)N, 0.0);
        const bool same_kind = (c->cost_kind == rt_type);
        // queue lengths: a source's own, whichever rank measured it (every source is traced by one rank, so the sum over the ranks
        // is the value and an unsummed array holds the value or nothing); a source run as sub-tree tasks keeps its last one
        if (c->src_q_h.size() != (size_t)N || (launching && !same_kind)) { c->src_q_h.assign((size_t)N, 0.0); c->src_q_est_h.assign((size_t)N, 0.0); }   // (a query, drt_get_sources, forgets nothing)
        if (same_kind) for (size_t i = 0; i < (size_t)N; i++) if (meas[(size_t)N + i] > 0) c->src_q_h[i] = meas[(size_t)N + i];
        double sum_meas = 0, sum_prior = 0;
        std::vector<double> lum(list.size());
        for (size_t k = 0; k < list.size(); k++) {
            const size_t i = (size_t)list[k];
            double s = 0;
            for (int il = 0; il < L; il++) s += c->lumcell_h[i * L + il];
            lum[k] = s;
            if (same_kind && meas[i] > 0 && c->cost_lum_h[i] > 0) { sum_meas += meas[i]; sum_prior += sqrt(c->cost_lum_h[i]); }
        }
        const double scale = sum_prior > 0 ? sum_meas / sum_prior : 1.0;
        // The deal among the ranks must come out the same on every rank: measured costs enter it only once they have been
        // summed over the ranks (drt_rt_mpi, or the host's own reduction followed by drt_mark_cost_reduced); until then a
        // rank knows the costs of its own sources only and the deal uses the prior, a function of lumcell, alone.
        const bool use_meas = same_kind && (np == 1 || c->cost_synced);
        c->pred_measured = use_meas && sum_meas > 0;
        for (size_t k = 0; k < list.size(); k++) {
            const size_t i = (size_t)list[k];
            if (use_meas && meas[i] > 0 && c->cost_lum_h[i] > 0) c->src_pred_h[i] = meas[i] * (lum[k] / c->cost_lum_h[i]);
            else c->src_pred_h[i] = (use_meas ? scale : 1.0) * sqrt(lum[k]);
            if (launching) c->cost_lum_h[i] = lum[k];  // the launch being prepared is the next one measured
        }
        if (launching) { c->cost_kind = rt_type; c->cost_synced = false; }
    }
    if (!iobs && !psrc && !by_mod && np > 1) {
        const size_t n = list.size();
        std::vector<int> mine;
        double csum = 0;
        for (int i : list) csum += c->src_pred_h[(size_t)i];
        const bool by_cost = !by_block && !(pe && !strcmp(pe, "cyclic")) && csum > 0;
        if (by_cost) {
            std::vector<double> cost(n);
            for (size_t k = 0; k < n; k++) cost[k] = c->src_pred_h[(size_t)list[k]];
            std::vector<int> owner;
            deal_by_cost(cost.data(), n, np, chunk, owner);
            for (size_t k = 0; k < n; k++) if (owner[k] == id) mine.push_back(list[k]);
        } else if (by_block) {
            const size_t lo = n * (size_t)id / np, hi = n * (size_t)(id + 1) / np;
            mine.assign(list.begin() + lo, list.begin() + hi);
        } else {
            for (size_t k0 = (size_t)id * chunk; k0 < n; k0 += (size_t)np * chunk)
                mine.insert(mine.end(), list.begin() + k0, list.begin() + std::min(n, k0 + chunk));
        }
        list.swap(mine);
    }
    if (list.size() > c->src_list_cap || !c->d_src_list) {
        if (c->d_src_list) (void)hipFree(c->d_src_list);
        c->src_list_cap = list.size() + 1024;
        HIPCHK(c, hipMalloc((void**)&c->d_src_list, c->src_list_cap * sizeof(int)));
    }
    if (!list.empty()) HIPCHK(c, hipMemcpy(c->d_src_list, list.data(), list.size() * sizeof(int), hipMemcpyHostToDevice));
    return DRT_OK;
}

static int check_device_error(drt_ctx* c)
{
    int e = 0;
    HIPCHK(c, hipMemcpy(&e, c->d_error, sizeof(int), hipMemcpyDeviceToHost));
    if (e) {
        (void)hipMemset(c->d_error, 0, sizeof(int));
        return fail(c, (e & (ERR_CAPACITY | ERR_LOG)) ? DRT_E_CAPACITY : DRT_E_TRACE, "device: %s (flags 0x%x)", trace_error_text(e), e);
    }
    return DRT_OK;
}

static int pull_counters(drt_ctx* c)
{
    unsigned long long h[CNT_N];
    HIPCHK(c, hipMemcpy(h, c->d_counters, sizeof h, hipMemcpyDeviceToHost));
    c->cnt.intersections = (int64_t)h[CNT_INTERSECTIONS]; c->cnt.rays = (int64_t)h[CNT_RAYS];
    c->cnt.deposits = (int64_t)h[CNT_DEPOSITS]; c->cnt.scaspe_adds = (int64_t)h[CNT_SCASPE];
    c->cnt.lambda_deposits = (int64_t)h[CNT_LAMDEP]; c->cnt.u_adds = (int64_t)h[CNT_UADD];
    c->cnt.u_reads = (int64_t)h[CNT_UREAD]; c->cnt.scaspe_slots = (int64_t)h[CNT_SCASLOTS];
    c->cnt.atomic_instructions = (int64_t)h[CNT_ATOM]; c->cnt.atomic_lanes = (int64_t)h[CNT_ATOM_LANES];
    return DRT_OK;
}

template <int LP> static void launch_stage(drt_ctx* c, const StageArgs& a, int blocks)
{
    hipLaunchKernelGGL(rt_stage_kernel<LP>, dim3(blocks), dim3(256), 0, 0, c->d_prob, a);
}
// the two-waves-per-SIMD instantiation of the one-wavelength kernel (cheap sources); at most what stays resident is launched
static void launch_wide_lp1w2(drt_ctx* c, const StageArgs& a, int blocks)
{
    const size_t lds = wide_lds_bytes(c);
    static int per_cu[3] = {0, 0, 0};
    const int m = a.log ? 2 : (c->big_tables ? 1 : 0);
    if (!per_cu[m]) {
        int nb = 0;
        hipError_t e = (m == 2) ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rt_stage_wide_lp1w2<2>, 256, lds)
                     : (m == 1) ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rt_stage_wide_lp1w2<1>, 256, lds)
                                : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rt_stage_wide_lp1w2<0>, 256, lds);
        per_cu[m] = (e == hipSuccess && nb > 0) ? nb : 1;
        if (getenv("DRT_TIMING")) fprintf(stderr, "[drt] rt_stage_wide_lp1w2: %d blocks per CU (occupancy query: %s)\n", per_cu[m], hipGetErrorString(e));
    }
    blocks = std::max(1, std::min(blocks, per_cu[m] * c->ncu));
    if (m == 2) hipLaunchKernelGGL((rt_stage_wide_lp1w2<2>), dim3(blocks), dim3(256), lds, 0, c->d_prob, a);
    else if (m == 1) hipLaunchKernelGGL((rt_stage_wide_lp1w2<1>), dim3(blocks), dim3(256), lds, 0, c->d_prob, a);
    else hipLaunchKernelGGL((rt_stage_wide_lp1w2<0>), dim3(blocks), dim3(256), lds, 0, c->d_prob, a);
}
template <int LP> static void launch_wide(drt_ctx* c, const StageArgs& a, int blocks)
{
    // deposit log: the rows are added by k_log_reduce afterwards; otherwise slot tables with more slots than lanes (kp >= 2)
    // need the instantiation with the memory-table path
    if (a.log) hipLaunchKernelGGL((rt_stage_wide<LP, 2>), dim3(blocks), dim3(256), wide_lds_bytes(c), 0, c->d_prob, a);
    else if (c->big_tables) hipLaunchKernelGGL((rt_stage_wide<LP, 1>), dim3(blocks), dim3(256), wide_lds_bytes(c), 0, c->d_prob, a);
    else hipLaunchKernelGGL((rt_stage_wide<LP, 0>), dim3(blocks), dim3(256), wide_lds_bytes(c), 0, c->d_prob, a);
}
template <int LP> static void launch_iobs(drt_ctx* c, int rt, int nsrc, int blocks)
{
    hipLaunchKernelGGL(rt_iobs_kernel<LP>, dim3(blocks), dim3(256), 0, 0, c->d_prob, rt, c->d_src_list, nsrc);
}

// ---- deposit log (rt_wide.hpp MODE 2, rt_log.hpp) --------------------------------------------------------------------------
static int ensure_log(drt_ctx* c)
{
    if (c->d_log) return DRT_OK;
    size_t bytes = c->log_bytes ? c->log_bytes : (16ull << 30);
    if (const char* e = getenv("DRT_LOG_GB")) bytes = (size_t)std::max(1, atoi(e)) << 30;
    if (const char* e = getenv("DRT_LOG_MB")) bytes = (size_t)std::max(8, atoi(e)) << 20;   // tests: a log that forces batches
    // index words per chunk: a record is 16 bytes + 8 per wavelength it has room for; a chunk whose index fills first is left early
    c->log_chunk_idx = c->log_chunk_bytes / (16u + 8u * (unsigned)std::max(1, c->L / 4));
    const size_t per_chunk = (size_t)c->log_chunk_bytes + 16ull * c->log_chunk_idx;  // records + index words + sorted offsets
    c->log_nchunks = (int)std::min<size_t>(bytes / per_chunk, 1u << 30);
    if (c->log_nchunks < 4) return fail(c, DRT_E_ARG, "deposit log of %zu bytes is too small", bytes);
    const size_t nidx = (size_t)c->log_nchunks * c->log_chunk_idx;
    HIPCHK(c, hipMalloc((void**)&c->d_log, (size_t)c->log_nchunks * c->log_chunk_bytes));
    HIPCHK(c, hipMalloc((void**)&c->d_log_idx, nidx * 8));
    HIPCHK(c, hipMalloc((void**)&c->d_log_sorted, nidx * 8));
    HIPCHK(c, hipMalloc((void**)&c->d_log_chunk_count, (size_t)c->log_nchunks * 4));
    HIPCHK(c, hipMalloc((void**)&c->d_log_cell_count, (size_t)c->N * 4));
    HIPCHK(c, hipMalloc((void**)&c->d_log_cursor, (size_t)c->N * 4));
    HIPCHK(c, hipMalloc((void**)&c->d_log_start, ((size_t)c->N + 1) * 8));
    HIPCHK(c, hipMalloc((void**)&c->d_log_bsum, ((size_t)c->N / DRT_SCAN_BLOCK + 2) * 8));
    HIPCHK(c, hipMalloc((void**)&c->d_log_used, sizeof(int)));
    HIPCHK(c, hipMalloc((void**)&c->d_log_rcounter, sizeof(int)));
    HIPCHK(c, hipMemset(c->d_log_used, 0, sizeof(int)));
    HIPCHK(c, hipMemset(c->d_log_chunk_count, 0, (size_t)c->log_nchunks * 4));
    HIPCHK(c, hipMemset(c->d_log_cell_count, 0, (size_t)c->N * 4));
    return DRT_OK;
}

// The records the last launch left in the log are binned by cell and every cell's rows are added to scaspe; the log is
// empty afterwards. `used`: chunks the launch filled (returned for the batch planning of the next call).
static int log_reduce(drt_ctx* c, int* used_out, bool may_skip = false)
{
    int used = 0;
    HIPCHK(c, hipMemcpy(&used, c->d_log_used, sizeof(int), hipMemcpyDeviceToHost));
    used = std::min(used, c->log_nchunks);
    if (used_out) *used_out = used;
    const bool no_reduce = may_skip && getenv("DRT_LOG_NOREDUCE") != nullptr;  // timing aid: the records of a scattering iteration are dropped
    if (used > 0 && no_reduce) HIPCHK(c, hipMemsetAsync(c->d_log_chunk_count, 0, (size_t)used * 4, 0));
    if (used > 0 && !no_reduce) {
        const int N = c->N, nb = (N + DRT_SCAN_BLOCK - 1) / DRT_SCAN_BLOCK;
        // (the records per cell were counted by the tracing kernel as it wrote them)
        HIPCHK(c, hipMemsetAsync(c->d_log_cursor, 0, (size_t)N * 4, 0));
        const dim3 g((unsigned)used, (unsigned)std::max(1u, c->log_chunk_idx / 1024u));
        hipLaunchKernelGGL(k_log_scan_sums, dim3(nb), dim3(256), 0, 0, c->d_log_cell_count, N, c->d_log_bsum);
        hipLaunchKernelGGL(k_log_scan_top, dim3(1), dim3(64), 0, 0, c->d_log_bsum, nb);
        hipLaunchKernelGGL(k_log_scan_fin, dim3(nb), dim3(64), 0, 0, c->d_log_cell_count, N, c->d_log_bsum, c->d_log_start, nb);
        hipLaunchKernelGGL(k_log_scatter, g, dim3(256), 0, 0, c->d_log_idx, c->d_log_chunk_count, c->log_chunk_idx, c->d_log_start, c->d_log_cursor, c->d_log_sorted);
        HIPCHK(c, hipMemsetAsync(c->d_log_rcounter, 0, sizeof(int), 0));
        const int blocks = std::min(N, c->ncu * 16);
        {
            const char* e = getenv("DRT_LOG_LG");   // wavelengths per work item of the reduce pass (developer knob)
            const int lg = e ? atoi(e) : 4;
            if (lg == 2) hipLaunchKernelGGL(k_log_reduce16<2>, dim3(blocks), dim3(64), 0, 0, c->d_prob, c->d_log, c->d_log_start, c->d_log_sorted, N, (c->L + 1) / 2, c->d_log_rcounter);
            else if (lg == 8) hipLaunchKernelGGL(k_log_reduce16<8>, dim3(blocks), dim3(64), 0, 0, c->d_prob, c->d_log, c->d_log_start, c->d_log_sorted, N, (c->L + 7) / 8, c->d_log_rcounter);
            else hipLaunchKernelGGL(k_log_reduce16<4>, dim3(blocks), dim3(64), 0, 0, c->d_prob, c->d_log, c->d_log_start, c->d_log_sorted, N, (c->L + 3) / 4, c->d_log_rcounter);
        }
        HIPCHK(c, hipMemsetAsync(c->d_log_chunk_count, 0, (size_t)used * 4, 0));
    }
    HIPCHK(c, hipMemsetAsync(c->d_log_cell_count, 0, (size_t)c->N * 4, 0));
    HIPCHK(c, hipMemsetAsync(c->d_log_used, 0, sizeof(int), 0));
    HIPCHK(c, hipGetLastError());
    return DRT_OK;
}

extern "C" int drt_rt_loop(drt_ctx* c, int32_t rt_type)
{
    if (!c) return DRT_E_ARG;
    if (!c->prepared) return fail(c, DRT_E_ARG, "drt_rt_loop before drt_rt_prepare(start)");
    const bool precalc = (rt_type == DRT_RTT_PRECALC_CELL || rt_type == DRT_RTT_PRECALC_SRC);
    const bool direct = (rt_type == DRT_RTT_DIR_CELL || rt_type == DRT_RTT_DIR_SRC || rt_type == DRT_RTT_SCATT);
    if (!precalc && !direct) return fail(c, DRT_E_ARG, "drt_rt_loop: rt_type %d is not a ray-tracing stage", rt_type);
    if (direct && !c->part2) return fail(c, DRT_E_ARG, "drt_rt_loop(%d) before drt_rt_prepare(prep_part2)", rt_type);
    if (rt_type == DRT_RTT_SCATT && c->iterations < 1) return fail(c, DRT_E_ARG, "drt_rt_loop(scatt) before drt_rt_prepare(scatt)");
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    std::vector<int> list;
    if ((rc = build_src_list(c, rt_type, list, true))) return rc;
    const int64_t inter_before = c->cnt.intersections;  // the device counters run on until drt_reset_counters
    c->cnt.sources = (int64_t)list.size();
    c->cnt.kernel_ms = 0; c->cnt.launches = 0; c->cnt.heavy_tasks = 0; c->cnt.deep_sources = 0; c->cnt.retries = 0;
    const bool cell_stage = !(rt_type == DRT_RTT_PRECALC_SRC || rt_type == DRT_RTT_DIR_SRC);
    if (cell_stage) HIPCHK(c, hipMemset(c->d_src_cost, 0, 2 * (size_t)c->N * sizeof(double)));  // build_src_list has read the previous launch's
    // psel_av: a cell stage starts from zero; the point-source stage that follows it writes its own entries [N, N + nsrc) beside the
    // cells' (one rt_mpi reduces both, psel_av_arr(.., iterations, sid) has one row for both, rt_routines.f90:1460). Done BEFORE the
    // empty-list return: a rank without sources of its own (fewer emitting cells than ranks) still takes part in rt_mpi's
    // reduction of these arrays, with zeros and not with the previous stage's sums.
    if (c->psel_av) {
        if (!c->ray_major) return fail(c, DRT_E_ARG, "psel_av is kept by the ray-major kernel only (DRT_RAY_MAJOR=0 is set)");
        const size_t M = (size_t)c->N + c->nsrc;
        if (cell_stage || !c->d_psel_sum) {
            if ((rc = dev_zero(c, &c->d_psel_sum, M))) return rc;
            if ((rc = dev_zero(c, &c->d_psel_cnt, M))) return rc;
        }
    }
    if (list.empty()) return DRT_OK;
    if ((rc = ensure_arena(c))) return rc;
    if (c->hprob.arena != c->d_arena || c->hprob.bm_par != c->bm_par_cur) if ((rc = upload_problem(c))) return rc;
    StageArgs a;
    a.rt_type = rt_type; a.nsources = (int)list.size(); a.src_list = c->d_src_list; a.scatter = direct ? 1 : 0;
    const bool psrc = (rt_type == DRT_RTT_PRECALC_SRC || rt_type == DRT_RTT_DIR_SRC);
    int G = (64 / (c->ray_major ? 1 : c->LP)) / 4;  // 4 children per sector: one task fills the wave's ray slots on its first step
    if (G < 1) G = 1;
    if (G > 16) G = 16;
    if (psrc) G = 1;  // few, deep sources: one sector per wave
    a.sectors_per_task = G;
    long long ntasks = (long long)list.size() * (48 / G);
    int blocks = (int)((ntasks + 3) / 4);
    if (blocks > c->waves / 4) blocks = c->waves / 4;
    a.export_level = 0; a.spawn_count = 0; a.spawn = nullptr; a.spawn_cap = 0; a.spawn_counter = nullptr;
    a.task_first = 0; a.task_stride = 1;
    a.heavy = nullptr; a.n_heavy = 0; a.record_cost = cell_stage ? 1 : 0;
    // Deposit log (rt_wide.hpp MODE 2): the stages that scatter leave records, the rows are added per cell afterwards.
    // DRT_LOG=1 switches it on (ray-major kernel only).
    a.log = nullptr; a.log_idx = nullptr; a.log_chunk_count = nullptr; a.log_chunks_used = nullptr; a.log_cell_count = nullptr;
    a.log_chunk_bytes = 0; a.log_chunk_idx = 0; a.log_nchunks = 0;
    {
        const char* e = getenv("DRT_LOG");
        c->use_log = e && e[0] == '1' && c->ray_major && c->N < (1 << 27);   // a record's key has 27 bits for the cell
    }
    const bool use_log = c->use_log && direct;
    if (use_log) {
        if ((rc = ensure_log(c))) return rc;
        a.log = c->d_log; a.log_idx = c->d_log_idx; a.log_chunk_count = c->d_log_chunk_count; a.log_chunks_used = c->d_log_used; a.log_cell_count = c->d_log_cell_count;
        a.log_chunk_bytes = c->log_chunk_bytes; a.log_chunk_idx = c->log_chunk_idx; a.log_nchunks = c->log_nchunks;
    }
    // batches: a stage whose records do not fit the log at once is cut into `nbatch` launches, every one followed by its
    // reduce pass; batch b takes the tasks b, b + nbatch, ... (an even mix of heavy and light ones). Planned from the chunks
    // the same stage filled last time; a launch that still overflows the log is undone and the stage repeated with twice as many.
    int nbatch = 1;
    if (use_log) {
        // from the fullest batch of the stage's last call: the batches are not equally full (the heaviest tasks come first in the
        // task list), and a batch that overflows costs an undone attempt and twice the batches
        const int peak = c->log_peak_last[rt_type & 15], nb_last = c->log_batches[rt_type & 15];
        if (peak > 0 && nb_last > 0) nbatch = (int)std::max(1.0, ceil((double)nb_last * (double)peak / (0.65 * (double)c->log_nchunks)));   // two thirds full: the fill of a batch varies by 10-20 % from call to call
        else if (nb_last > 0) nbatch = nb_last;
        if (const char* e = getenv("DRT_LOG_BATCHES")) nbatch = std::max(1, atoi(e));
    }
    long long log_chunks_total = 0;
    int log_peak = 0;
    bool ovf_log = false;
    int g_used = 0;   // sectors per task of the cell launch that went through
    double f_used_stage = 0.0;   // and its task limit as a fraction of a wave's share
    double qmax_stage = 0.0;   // the longest ray queue of the cell launch that went through
    a.psel_sum = nullptr; a.psel_cnt = nullptr;
    if (c->psel_av) { a.psel_sum = c->d_psel_sum; a.psel_cnt = c->d_psel_cnt; }
    // Phase window (trace_items_w): 1.5 cell sizes where deposits dominate; off where a source launches few rays (the
    // work per source of the previous call of this stage, like the sectors per task) and the emptier steps cost more
    // than the merged atomics save. DRT_PHASE_WINDOW overrides.
    // (one wavelength, bench-sized grid: 0 -> 713 ms, 1 -> 848, 2 -> 753: not bound by the atomic unit; 32 wavelengths: 6.62 -> 5.99 s)
    // (with u_final batched per chunk, bench step: 0.5 -> 5.18 s, 0.75 -> 4.68, 1 -> 4.43, 1.25 -> 4.35, 1.5 -> 4.35, 2 -> 4.56, 2.5 -> 4.80, 3 -> 4.90)
    // (one wavelength, bench-sized grid, after this round's register diet: 0 -> 534 ms, 0.75 -> 585, 1 -> 536, 1.25 -> 511, 1.5 -> 503, 2 -> 513, 3 -> 535;
    //  config 1, cheap sources: 0 -> 4.9 ms, 1.5 -> 5.2)
    a.phase_win = c->cheap_sources[rt_type & 15] ? 0.0 : 1.5;
    if (const char* e = getenv("DRT_PHASE_WINDOW")) a.phase_win = atof(e);
    auto launch = [&](int nblocks) {
        if (c->ray_major) {
            switch (c->LP) {
            case 1:
                if (c->cheap_sources[rt_type & 15] && !psrc && !getenv("DRT_LP1_THREE_WAVES")) launch_wide_lp1w2(c, a, nblocks);
                else launch_wide<1>(c, a, nblocks);
                break;
            case 2: launch_wide<2>(c, a, nblocks); break;
            case 4: launch_wide<4>(c, a, nblocks); break;
            case 8: launch_wide<8>(c, a, nblocks); break;
            case 16: launch_wide<16>(c, a, nblocks); break;
            case 32: launch_wide<32>(c, a, nblocks); break;
            default: launch_wide<64>(c, a, nblocks); break;
            }
            return;
        }
        switch (c->LP) {
        case 1:
            if (lp1_three_waves(c)) hipLaunchKernelGGL(rt_stage_kernel_1w3, dim3(nblocks), dim3(256), 0, 0, c->d_prob, a);
            else launch_stage<1>(c, a, nblocks);
            break;
        case 2: launch_stage<2>(c, a, nblocks); break;
        case 4: launch_stage<4>(c, a, nblocks); break;
        case 8: launch_stage<8>(c, a, nblocks); break;
        case 16: launch_stage<16>(c, a, nblocks); break;
        case 32: launch_stage<32>(c, a, nblocks); break;
        default: launch_stage<64>(c, a, nblocks); break;
        }
    };
    // The accumulators this stage adds to are snapshotted (device-to-device, ~0.4 ms per GB): when a ray queue outgrows
    // its wave's arena slice every wave stops at its next task (they poll the error word), the launch is undone and
    // repeated with deeper queues - half the sectors per task for cell sources, half the waves for point sources.
    const size_t NL = (size_t)c->N * c->L;
    if (!c->d_bak_u) HIPCHK(c, hipMalloc((void**)&c->d_bak_u, NL * sizeof(double)));
    if (!c->d_bak_lost) HIPCHK(c, hipMalloc((void**)&c->d_bak_lost, (size_t)c->L * sizeof(double)));
    if (!c->d_bak_cnt) HIPCHK(c, hipMalloc((void**)&c->d_bak_cnt, CNT_N * sizeof(unsigned long long)));
    if (direct && c->d_bak_scaspe.size() != (size_t)c->L) {
        for (auto p : c->d_bak_scaspe) if (p) (void)hipFree(p);
        c->d_bak_scaspe.assign(c->L, nullptr);
        for (int il = 0; il < c->L; il++) HIPCHK(c, hipMalloc((void**)&c->d_bak_scaspe[il], (size_t)c->N * c->nstride[il] * sizeof(double)));
    }
    double* acc_u = precalc ? c->d_u_fest : c->d_u_final;
    auto copy_state = [&](bool save) -> int {
        auto cp = [&](void* bak, void* live, size_t bytes) { return save ? hipMemcpyAsync(bak, live, bytes, hipMemcpyDeviceToDevice, 0) : hipMemcpyAsync(live, bak, bytes, hipMemcpyDeviceToDevice, 0); };
        HIPCHK(c, cp(c->d_bak_u, acc_u, NL * sizeof(double)));
        HIPCHK(c, cp(c->d_bak_cnt, c->d_counters, CNT_N * sizeof(unsigned long long)));
        if (a.psel_sum && !psrc) {   // a cell stage's sums and counts of ended rays (the point-source stage clears its own entries per attempt)
            const size_t M = (size_t)c->N + c->nsrc;
            if (c->bak_psel_n < 2 * M) { if (c->d_bak_psel) (void)hipFree(c->d_bak_psel); c->d_bak_psel = nullptr; HIPCHK(c, hipMalloc((void**)&c->d_bak_psel, 2 * M * sizeof(double))); c->bak_psel_n = 2 * M; }
            HIPCHK(c, cp(c->d_bak_psel, c->d_psel_sum, M * sizeof(double)));
            HIPCHK(c, cp(c->d_bak_psel + M, c->d_psel_cnt, M * sizeof(double)));
        }
        if (direct) {
            HIPCHK(c, cp(c->d_bak_lost, c->d_lum_lost, (size_t)c->L * sizeof(double)));
            for (int il = 0; il < c->L; il++) HIPCHK(c, cp(c->d_bak_scaspe[il], c->d_scaspe[il], (size_t)c->N * c->nstride[il] * sizeof(double)));
        }
        return DRT_OK;
    };
    // device error word after a launch: 0, "queue overflow only" (recoverable: reset and report true), or something else
    auto overflowed = [&](bool& ovf) -> int {
        int e = 0;
        HIPCHK(c, hipMemcpy(&e, c->d_error, sizeof(int), hipMemcpyDeviceToHost));
        ovf = (e & (ERR_CAPACITY | ERR_LOG)) && !(e & ~(ERR_CAPACITY | ERR_LOG));
        ovf_log = ovf && (e & ERR_LOG) && !(e & ERR_CAPACITY);  // only the log was too small: more batches, same queues
        if (ovf) HIPCHK(c, hipMemset(c->d_error, 0, sizeof(int)));
        return DRT_OK;
    };
    // one launch of the stage kernel over the tasks a.task_first + k * a.task_stride - or, with the deposit log, nbatch
    // launches over disjoint shares of them, each followed by its reduce pass
    int nbatch_sub = 0;   // batches of the sub-tree launches of a cell stage (0: as many as the stage's own)
    bool in_subtree = false;
    auto run_batches = [&](int nblocks, bool& ovf) -> int {
        const int B = use_log ? ((in_subtree && nbatch_sub > 0) ? nbatch_sub : nbatch) : 1;
        const int f0 = a.task_first, s0 = a.task_stride;
        int rcb = DRT_OK;
        ovf = false;
        for (int b = 0; b < B && !ovf && !rcb; b++) {
            a.task_first = f0 + s0 * b; a.task_stride = s0 * B;
            rcb = (hipMemset(c->d_task, 0, sizeof(int)) == hipSuccess) ? DRT_OK : fail(c, DRT_E_HIP, "hipMemset(d_task)");
            if (rcb) break;
            // every wave that writes records holds a chunk of the log: with a small log fewer waves run (half the chunks at most)
            launch(use_log ? std::max(1, std::min(nblocks, c->log_nchunks / 8)) : nblocks);
            c->cnt.launches++;
            if ((rcb = overflowed(ovf))) break;
            if (use_log) {
                int used = 0;
                if (ovf) { (void)hipMemset(c->d_log_used, 0, sizeof(int)); (void)hipMemset(c->d_log_chunk_count, 0, (size_t)c->log_nchunks * 4); (void)hipMemset(c->d_log_cell_count, 0, (size_t)c->N * 4); }
                else if ((rcb = log_reduce(c, &used, rt_type == DRT_RTT_SCATT))) break;
                log_chunks_total += used;
                if (!in_subtree) log_peak = std::max(log_peak, used);
            }
        }
        a.task_first = f0; a.task_stride = s0;
        return rcb;
    };
    HIPCHK(c, hipEventRecord(c->ev0, 0));
    // an attempt that overflowed and is about to be undone: its device time stays in kernel_ms (ev0 is recorded again for the
    // next attempt), and it is counted
    double ms_undone = 0.0;
    auto attempt_failed = [&]() -> int {
        float ms_a = 0;
        HIPCHK(c, hipEventRecord(c->ev1, 0));
        HIPCHK(c, hipEventSynchronize(c->ev1));
        HIPCHK(c, hipEventElapsedTime(&ms_a, c->ev0, c->ev1));
        ms_undone += ms_a;
        c->cnt.retries++;
        return DRT_OK;
    };
    // Sub-tree tasks (StageArgs::export_level) for the `nsub` sources of `sub_list` (device): phase 1 traces their 192 nside-4
    // rays, one sector per task, and exports the blocked ones (at most 192 per source); phase 2 runs one task per child pixel
    // of an exported ray. `first`/`stride`: the phase-1 tasks this rank takes. ovf: a queue overflowed (state to be restored).
    auto subtree = [&](const int* sub_list, int nsub, int first, int stride, int max_blocks, bool& ovf, bool& retry_fewer) -> int {
        ovf = false; retry_fewer = false;
        const int need = 192 * nsub;
        if (need > c->spawn_cap || !c->d_spawn) {
            if (c->d_spawn) (void)hipFree(c->d_spawn);
            c->spawn_cap = need;
            HIPCHK(c, hipMalloc((void**)&c->d_spawn, list_bytes(need, c->LP) + 4ull * need));
        }
        if (!c->d_spawn_counter) HIPCHK(c, hipMalloc((void**)&c->d_spawn_counter, sizeof(int)));
        StageArgs keep = a;
        in_subtree = true;
        struct Leave { bool& f; ~Leave() { f = false; } } leave{in_subtree};
        a.src_list = sub_list; a.nsources = nsub; a.sectors_per_task = 1; a.heavy = nullptr; a.n_heavy = 0;
            a.spawn = c->d_spawn; a.spawn_cap = c->spawn_cap; a.spawn_counter = c->d_spawn_counter;
        HIPCHK(c, hipMemset(c->d_spawn_counter, 0, sizeof(int)));
        a.export_level = 4; a.spawn_count = 0;
        a.task_first = first; a.task_stride = stride;
        const long long nt1 = 48ll * nsub;
        const int blocks1 = (int)std::max(1ll, std::min((nt1 + 3) / 4, (long long)max_blocks));
        int rc2 = carve_arena(c, blocks1 * 4, a);
        if (rc2) {  // not even 64-ray queues for this many waves: use fewer
            a = keep;
            if (rc2 == DRT_E_CAPACITY && blocks1 > 1) { retry_fewer = true; return DRT_OK; }
            return rc2;
        }
        if ((rc2 = run_batches(blocks1, ovf))) { a = keep; return rc2; }
        int nspawn = 0;
        HIPCHK(c, hipMemcpy(&nspawn, c->d_spawn_counter, sizeof(int), hipMemcpyDeviceToHost));
        if (!ovf && nspawn > 0) {
            a.export_level = 0; a.spawn_count = std::min(nspawn, c->spawn_cap);
            a.task_first = 0; a.task_stride = 1;  // phase 2 runs what THIS rank exported
            const long long nt = 4ll * a.spawn_count;
            int blocks2 = (int)std::max(1ll, std::min((nt + 3) / 4, (long long)max_blocks));
            while ((rc2 = carve_arena(c, blocks2 * 4, a)) == DRT_E_CAPACITY && blocks2 > 1) blocks2 = (blocks2 + 1) / 2;
            if (rc2) { a = keep; return rc2; }
            if ((rc2 = run_batches(blocks2, ovf))) { a = keep; return rc2; }
        }
        a = keep;
        return DRT_OK;
    };
    if (psrc) {
        // few, deep sources; the sector tasks of phase 1 are dealt round the ranks
        for (int max_blocks = c->waves / 4;;) {
            if ((rc = copy_state(true))) return rc;
            if (a.psel_sum && c->nsrc > 0) {  // not in the snapshot: an undone attempt must not leave its rays in the sums
                HIPCHK(c, hipMemsetAsync(c->d_psel_sum + c->N, 0, (size_t)c->nsrc * sizeof(double), 0));
                HIPCHK(c, hipMemsetAsync(c->d_psel_cnt + c->N, 0, (size_t)c->nsrc * sizeof(double), 0));
            }
            HIPCHK(c, hipEventRecord(c->ev0, 0));
            bool ovf = false, fewer = false;
            if ((rc = subtree(c->d_src_list, (int)list.size(), c->par.id_mpi, c->par.np_mpi, max_blocks, ovf, fewer))) return rc;
            if (fewer) { max_blocks = std::max(1, std::min(max_blocks, (int)((48ll * (long long)list.size() + 3) / 4)) / 2); continue; }
            if (!ovf) break;
            if ((rc = attempt_failed())) return rc;
            if ((rc = copy_state(false))) return rc;  // undo the partial launch in any case
            if (ovf_log) { if (nbatch >= 4096) return fail(c, DRT_E_CAPACITY, "deposit log too small"); nbatch *= 2; log_chunks_total = 0; log_peak = 0; continue; }
            if (max_blocks <= 1) return fail(c, DRT_E_CAPACITY, "device: %s", trace_error_text(ERR_CAPACITY));
            max_blocks = (max_blocks + 1) / 2;
        }
    } else {
        // Cell sources: G sectors per task. More sectors fill the lanes better (and merge more rays per atomic) but
        // lengthen the queues; how long they get depends on the grid. The value that fitted is kept for the next call.
        // Sectors per task. 16 fill a wave's 64 lanes with first-level rays and merge most rays per atomic: best when a
        // source launches many rays (TRUST-I tau = 1: ~8e4 intersections per source) - and, since round 3, also when a source
        // is cheap (tau = 0.01: ~9e2; see where G_want is set). The choice follows the intersections per source of the
        // previous call of the same stage (a run repeats its scattering stage); first call: 8.
        int Gtry = c->G_want[rt_type & 15] > 0 ? c->G_want[rt_type & 15] : 8;
        if (const char* e = getenv("DRT_SECTORS_WANT")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) Gtry = v; }   // developer knob: overrides the choice by work per source
        // the sources' own queue lengths (src_q_h, plan_tasks) make both memories below unnecessary: each source is cut by its own
        // queue, so the stage goes for the sectors it wants (bench grid with the default 16 GiB arena: the first call had ended at one
        // sector per task on an eighth of the waves and every later call stayed at one sector, 4.77 s per step)
        size_t q_known = 0;
        if (!getenv("DRT_NO_SRC_QUEUES") && c->src_q_h.size() == (size_t)c->N) for (int i : list) q_known += c->src_q_h[(size_t)i] > 0;
        const bool own_queues = 2 * q_known >= list.size() && !list.empty();   // (build_src_list forgets them when the kind of stage changes)
        if (c->G_cell[rt_type & 15] > 0 && !own_queues) Gtry = std::min(Gtry, c->G_cell[rt_type & 15]);
        // more sectors per task than the stage's last launch took: only if its queues left the room (bench grid, 32 wavelengths:
        // the launch at 8 sectors fills 60-70 % of the queues of a 200 GiB arena; trying 16 meant an overflow, an undone attempt
        // of 3.4 s and a repeat at 8 - in the second call of every stage)
        if (c->G_last[rt_type & 15] > 0 && !getenv("DRT_SECTORS_WANT") && !own_queues)
            while (Gtry > c->G_last[rt_type & 15] && c->q_fill[rt_type & 15] * (double)Gtry / (double)c->G_last[rt_type & 15] > 0.85) Gtry /= 2;
        double f_used = 0.0;
        Gtry = std::min(Gtry, G);
        if (const char* e = getenv("DRT_SECTORS_PER_TASK")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) Gtry = std::min(Gtry, v); }
        int max_blocks = c->waves / 4;
        // A task is one wave's serial work, so no task may be a large part of the whole launch (TRUST-I, one wavelength,
        // 85 801 cells: a first-level cell costs 1 s of wave time in a 0.7 s launch, and with 8 ranks the launch is 0.09 s).
        // From the expected cost of every source (src_pred_h: the wave cycles it took in the previous launch of this kind of
        // stage, DRT_A_SRC_COST, scaled with its luminosity; sqrt(luminosity) for a first launch), with
        // limit = ideal launch time / DRT_HEAVY_DIV (default 8, 0 = off):
        //  * heavy sources - a task of Gtry sectors would exceed the limit - get fewer sectors per task, down to one; their
        //    tasks are listed explicitly (StageArgs::heavy) and go first, heaviest source first;
        //  * deep sources - even one sector (1/48 of the cost on average; the sectors in the plane of a slab cost several
        //    times that) would exceed it - are run as sub-tree tasks like the point sources, before the others.
        double heavy_div = 8.0;   // 16 until round 3: no difference on config 2 (4.25 s either way) and the one-wavelength step (504 ms), 2.06 against 2.38 ms on config 1
        if (const char* e = getenv("DRT_HEAVY_DIV")) heavy_div = atof(e);
        std::vector<int> light, deep;
        std::vector<int2> heavy;
        double stage_total = 0;       // sum of the expected costs of all the stage's sources when it runs in parts (else 0)
        std::vector<int> cur;         // the sources of the launch being prepared: all of the stage's, or one part of them (see the probe below)
        std::vector<double> q_plan;   // a source's measured queue length, else the estimate made after an undone launch, else 0
        auto plan = [&](int Gbase) {
            q_plan.assign((size_t)c->N, 0.0);
            if (c->src_q_h.size() == (size_t)c->N)
                for (int i : cur) q_plan[(size_t)i] = c->src_q_h[(size_t)i] > 0 ? c->src_q_h[(size_t)i] : (c->src_q_est_h.size() == (size_t)c->N ? c->src_q_est_h[(size_t)i] : 0.0);
            double deep_min = 4.0e6;   // shader-clock cycles of wave time per wave (~1.7 ms) below which no source is run as sub-tree tasks:
                                       // config 1 (a 2 ms step): 96 deep sources -> 3.37 ms per step, none -> 2.38, with the limit at 1/8
                                       // 2.06; a 5 ms threshold lost the third scattering iteration of an 8-rank run (36.7 against 15.9 ms)
            if (const char* e = getenv("DRT_DEEP_MIN")) deep_min = atof(e);
            // Unmeasured launches are cut four times finer: on the bench grid the first scattering iteration runs through at 8 sectors
            // (4.25 s) instead of overflowing its queues, being undone and repeated at 4 (5.81 s). The queue rule: the first launch,
            // cut at 1/32, fills 52 % of the queues; the second, measured, would be cut at 1/8, overflow and be repeated - at 1/20 it
            // goes through.
            PlanIn in{cur.data(), cur.size(), c->src_pred_h.data(), c->waves, Gbase, nbatch, heavy_div, deep_min, c->pred_measured,
                      getenv("DRT_NO_DEEP") != nullptr, c->f_last[rt_type & 15], c->q_fill[rt_type & 15], c->G_last[rt_type & 15],
                      getenv("DRT_NO_SRC_QUEUES") ? nullptr : q_plan.data(), (double)queue_cap(c, max_blocks * 4, Gbase), stage_total};
            PlanOut o;
            plan_tasks(in, o);
            light.swap(o.light); heavy.swap(o.heavy); deep.swap(o.deep);
            nbatch_sub = o.nbatch_sub;
            f_used = o.f_used;
            if (getenv("DRT_TIMING")) fprintf(stderr, "[drt] plan rt %d: total %.3e cycles, per wave %.3e, limit %.3e, sources %zu -> deep %zu heavy %zu light %zu (G %d; %zu cut by their queues, %d rays per queue)\n", rt_type, o.total, o.total / c->waves, o.limit, cur.size(), deep.size(), o.n_heavy_src, light.size(), Gbase, o.n_queue_cut, (int)in.cap);
        };
        // Queue lengths of the sources nothing was measured of, from the ones that were (fit_queue_prior): the line through
        // (log luminosity, log queue) times what 99.5 % of the measured sources stay below (at least 2, at most 8).
        auto estimate_queues = [&]() {
            if (getenv("DRT_NO_QUEUE_PRIOR")) return;
            std::vector<double> xl, yq;
            for (int i : list) if (c->src_q_h[(size_t)i] > 0) { xl.push_back(c->cost_lum_h[(size_t)i]); yq.push_back(c->src_q_h[(size_t)i]); }
            double fa = 0, fb = 0;
            if (!fit_queue_prior(xl.data(), yq.data(), xl.size(), fa, fb)) return;
            std::vector<double> res;
            for (size_t k = 0; k < xl.size(); k++) if (xl[k] > 0) res.push_back(yq[k] / exp(fa + fb * log(xl[k])));
            double R = 2.0;
            if (res.size() >= 64) { const size_t kq = (size_t)(0.995 * (double)(res.size() - 1)); std::nth_element(res.begin(), res.begin() + kq, res.end()); R = std::min(8.0, std::max(2.0, res[kq])); }
            if (c->src_q_est_h.size() != (size_t)c->N) c->src_q_est_h.assign((size_t)c->N, 0.0);
            for (int i : list) { const double lum = c->cost_lum_h[(size_t)i]; c->src_q_est_h[(size_t)i] = lum > 0 ? R * exp(fa + fb * log(lum)) : 0.0; }
            if (getenv("DRT_TIMING")) fprintf(stderr, "[drt] rt %d: queue prior from %zu sources: q48 = %.3g * luminosity^%.2f x %.2f\n", rt_type, xl.size(), exp(fa), fb, R);
        };
        // The probe: a stage of which no queue length is known runs a sixteenth of its sources first (every sixteenth chunk of 64
        // consecutive ones: neighbours stay together, the sample covers the model) and plans the others by what these showed -
        // their own queue lengths give the line and the spread (estimate_queues). An overflow in the probe undoes a sixteenth of
        // the stage at most; without it the first overflow of the N-body direct light came 7 s into a 15.7 s stage. Not for
        // fewer than 8192 sources: with an eighth of the sources and from 4096 on, the first scattering iteration of an 8-rank run
        // of the bench grid (7475 sources and 0.6 s per rank, no overflow to avoid) took 9 % longer.
        std::vector<std::vector<int>> parts;
        {
            size_t q_any = 0;
            if (c->src_q_h.size() == (size_t)c->N) for (int i : list) q_any += c->src_q_h[(size_t)i] > 0;
            const bool probe = q_any == 0 && list.size() >= 8192 && !use_log && !getenv("DRT_NO_PROBE") && !getenv("DRT_NO_SRC_QUEUES") && c->ray_major;
            if (probe) {
                parts.resize(2);
                for (size_t k = 0; k < list.size(); k++) parts[((k >> 6) & 15) == 0 ? 0 : 1].push_back(list[k]);
                for (int i : list) stage_total += c->src_pred_h[(size_t)i];   // a task may be as large as in the whole stage's launch: the probe is not cut eight times finer
            } else parts.push_back(list);
        }
        std::vector<double> cost_commit(2 * (size_t)c->N, 0.0);   // src_cost as of the last part that went through
        if (a.psel_sum) {
            HIPCHK(c, hipMemsetAsync(c->d_psel_sum, 0, ((size_t)c->N + c->nsrc) * sizeof(double), 0));
            HIPCHK(c, hipMemsetAsync(c->d_psel_cnt, 0, ((size_t)c->N + c->nsrc) * sizeof(double), 0));
        }
        int64_t heavy_done = 0, deep_done = 0;
        bool kept_once = false;
        // The kernel leaves, per source, the longest queue among its whole-sector tasks (second half of src_cost). All tasks of a
        // source take the same number of sectors - the plan's - so the host turns the array into "queue at 48 sectors per task",
        // the form the planner, the other ranks (the array is summed over the ranks with the costs) and the next call use.
        // Returns the longest queue itself.
        auto scale_queues = [&](int Gl, double& qmax) -> int {
            qmax = 0.0;
            if (heavy.size() + light.size() + deep.size() == 0) return DRT_OK;
            std::vector<double> qh((size_t)c->N);
            HIPCHK(c, hipMemcpy(qh.data(), c->d_src_cost + c->N, (size_t)c->N * sizeof(double), hipMemcpyDeviceToHost));
            for (int i : deep) qh[(size_t)i] = 0.0;   // sub-tree tasks say nothing about whole sectors: these sources keep what is known of them
            for (int i : light) { qmax = std::max(qmax, qh[(size_t)i]); qh[(size_t)i] *= 48.0 / Gl; }
            int last = -1;
            for (const int2& h : heavy) if (h.x != last) { last = h.x; qmax = std::max(qmax, qh[(size_t)h.x]); qh[(size_t)h.x] *= 48.0 / (h.y >> 8); }   // a source's tasks are listed together
            HIPCHK(c, hipMemcpy(c->d_src_cost + c->N, qh.data(), (size_t)c->N * sizeof(double), hipMemcpyHostToDevice));
            return DRT_OK;
        };
        for (size_t part = 0; part < parts.size(); part++) {
        cur = parts[part];
        kept_once = false;
        // the probe runs ONE sector per task: 48 times the longest queue of a single sector bounds the source's queue at any number of
        // sectors, so what the others are planned by errs on the safe side (measured at 8 sectors the spread round the line - up
        // to x 22 on TRUST-I - let the rest of the N-body direct light overflow twice: 25.5 s; this way not at all: 16.1 s)
        const int G_stage = Gtry, blocks_stage = max_blocks;
        if (parts.size() > 1 && part == 0) Gtry = 1;
        for (;;) {
            plan(Gtry);
            if (heavy.size() > c->heavy_cap || !c->d_heavy) {
                if (c->d_heavy) (void)hipFree(c->d_heavy);
                c->heavy_cap = heavy.size() + 4096;
                HIPCHK(c, hipMalloc((void**)&c->d_heavy, c->heavy_cap * sizeof(int2)));
            }
            if (deep.size() > c->deep_cap || !c->d_deep) {
                if (c->d_deep) (void)hipFree(c->d_deep);
                c->deep_cap = deep.size() + 1024;
                HIPCHK(c, hipMalloc((void**)&c->d_deep, c->deep_cap * sizeof(int)));
            }
            if (!heavy.empty()) HIPCHK(c, hipMemcpy(c->d_heavy, heavy.data(), heavy.size() * sizeof(int2), hipMemcpyHostToDevice));
            if (!deep.empty()) HIPCHK(c, hipMemcpy(c->d_deep, deep.data(), deep.size() * sizeof(int), hipMemcpyHostToDevice));
            if (!light.empty()) HIPCHK(c, hipMemcpy(c->d_src_list, light.data(), light.size() * sizeof(int), hipMemcpyHostToDevice));
            c->cnt.heavy_tasks = heavy_done + (int64_t)heavy.size(); c->cnt.deep_sources = deep_done + (int64_t)deep.size();
            if ((rc = copy_state(true))) return rc;
            HIPCHK(c, hipMemcpy(c->d_src_cost, cost_commit.data(), cost_commit.size() * sizeof(double), hipMemcpyHostToDevice));   // what an undone attempt left is gone
            HIPCHK(c, hipEventRecord(c->ev0, 0));
            bool ovf = false, fewer = false, ovf_sub = false;
            if (!deep.empty()) {
                if ((rc = subtree(c->d_deep, (int)deep.size(), 0, 1, max_blocks, ovf, fewer))) return rc;
                if (fewer) { max_blocks = std::max(1, max_blocks / 2); continue; }
                ovf_sub = ovf;
            }
            if (!ovf && heavy.size() + light.size() > 0) {
                a.sectors_per_task = Gtry;
                a.heavy = c->d_heavy; a.n_heavy = (int)heavy.size(); a.nsources = (int)light.size();
                ntasks = (long long)heavy.size() + (long long)light.size() * (48 / Gtry);
                blocks = (int)std::max(1ll, std::min((ntasks + 3) / 4, (long long)max_blocks));
                if ((rc = carve_arena(c, blocks * 4, a))) {  // not even 64-ray queues for this many waves: use fewer
                    if (rc == DRT_E_CAPACITY && blocks > 1) { if ((rc = copy_state(false))) return rc; max_blocks = (blocks + 1) / 2; continue; }
                    return rc;
                }
                if ((rc = run_batches(blocks, ovf))) return rc;
            }
            if (!ovf) break;
            if ((rc = attempt_failed())) return rc;
            if ((rc = copy_state(false))) return rc;  // undo the partial launch in any case
            if (ovf_log) { if (nbatch >= 4096) return fail(c, DRT_E_CAPACITY, "deposit log too small"); nbatch *= 2; log_chunks_total = 0; log_peak = 0; continue; }
            if (ovf_sub) {
                // the sub-tree tasks of the deep sources overflowed: they run one sector (then one nside-8 pixel) per task
                // whatever Gtry is, so only deeper queues help - fewer waves. The light tasks never ran: neither Gtry nor the
                // remembered G_cell has anything to learn from this.
                if (max_blocks > 1) max_blocks = (max_blocks + 1) / 2;
                else return fail(c, DRT_E_CAPACITY, "device: %s", trace_error_text(ERR_CAPACITY));
                continue;
            }
            c->overflow_met = true;
            // what the undone launch learnt about queues: the tasks that were running when it stopped finished and left their sources'
            // queue lengths, the ones that overflowed a bound (rt_stage_wide). The repeat cuts those sources by them; the first
            // repeat of a call keeps the sectors per task for all the others.
            bool learnt = false;
            if (!getenv("DRT_NO_SRC_QUEUES") && c->src_q_h.size() == (size_t)c->N) {
                double qm;
                if ((rc = scale_queues(Gtry, qm))) return rc;
                std::vector<double> qh((size_t)c->N);
                HIPCHK(c, hipMemcpy(qh.data(), c->d_src_cost + c->N, (size_t)c->N * sizeof(double), hipMemcpyDeviceToHost));
                for (size_t i = 0; i < (size_t)c->N; i++) if (qh[i] > c->src_q_h[i]) { c->src_q_h[i] = qh[i]; learnt = true; }
                // ... and about the sources that had not run yet: twice what the line through (luminosity, queue) of the measured
                // ones gives (N-body direct light with the default arena: four undone attempts of ~3 s each, every one stopped by
                // a source the one before had not reached)
                estimate_queues();
            }
            if (learnt && !kept_once) { kept_once = true; continue; }
            if (Gtry > 1) Gtry /= 2;                  // shorter queues first,
            else if (max_blocks > 1) max_blocks = (max_blocks + 1) / 2;  // then fewer waves with deeper ones
            else return fail(c, DRT_E_CAPACITY, "device: %s", trace_error_text(ERR_CAPACITY));
        }
        // the part went through
        {
            double qm = 0.0;
            if ((rc = scale_queues(Gtry, qm))) return rc;
            qmax_stage = std::max(qmax_stage, qm);
            HIPCHK(c, hipMemcpy(cost_commit.data(), c->d_src_cost, cost_commit.size() * sizeof(double), hipMemcpyDeviceToHost));
            heavy_done += (int64_t)heavy.size(); deep_done += (int64_t)deep.size();
            if (part + 1 < parts.size()) {
                // the probe's queue lengths are known now, not at the next call: they make the estimate for the other sources
                for (int i : cur) if (cost_commit[(size_t)c->N + (size_t)i] > 0) c->src_q_h[(size_t)i] = cost_commit[(size_t)c->N + (size_t)i];
                estimate_queues();
                float ms_p = 0;   // its time stays in kernel_ms (ev0 is recorded again for the next part)
                HIPCHK(c, hipEventRecord(c->ev1, 0));
                HIPCHK(c, hipEventSynchronize(c->ev1));
                HIPCHK(c, hipEventElapsedTime(&ms_p, c->ev0, c->ev1));
                ms_undone += ms_p;
                Gtry = G_stage; max_blocks = blocks_stage;   // what the probe had to give up (sectors it never had; waves, if its queues overflowed twice) is the probe's affair
            }
        }
        }
        if (c->overflow_met) c->G_cell[rt_type & 15] = c->G_cell[rt_type & 15] > 0 ? std::min(c->G_cell[rt_type & 15], Gtry) : Gtry;  // an overflow was met: more sectors than this do not fit this arena
        g_used = Gtry; f_used_stage = f_used;
        c->overflow_met = false;
    }
    HIPCHK(c, hipGetLastError());
    if (use_log) { c->log_batches[rt_type & 15] = nbatch; c->log_chunks_last[rt_type & 15] = log_chunks_total; c->log_peak_last[rt_type & 15] = log_peak; }
    if (direct && c->ray_major && c->LP > 1 && !use_log)
        for (int il = 0; il < c->L; il++)
            if (c->npix[il] < c->nstride[il] && c->npix[il] < 64)
                hipLaunchKernelGGL(k_fold_pad, dim3((c->N + 255) / 256), dim3(256), 0, 0, c->d_scaspe[il], c->d_u_final, c->N, c->L, il, c->npix[il], c->nstride[il]);
    HIPCHK(c, hipEventRecord(c->ev1, 0));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->cnt.kernel_ms = ms + ms_undone;
    if ((rc = check_device_error(c))) return rc;
    if ((rc = pull_counters(c))) return rc;
    if (!psrc && g_used > 0 && c->cap > 0) {
        const unsigned long long qmax = (unsigned long long)(qmax_stage + 0.5);
        c->G_last[rt_type & 15] = g_used;
        c->q_fill[rt_type & 15] = (double)qmax / (double)c->cap;
        c->f_last[rt_type & 15] = f_used_stage;
        if (getenv("DRT_TIMING")) fprintf(stderr, "[drt] rt %d: %d sectors per task, longest queue %llu of %d rays\n", rt_type, g_used, qmax, c->cap);
    }
    if (!psrc && c->cnt.sources > 0) {
        const double ips = (double)(c->cnt.intersections - inter_before) / (double)c->cnt.sources;
        // Cheap sources (TRUST-I tau = 0.01: ~9e2 intersections per source) are a handful of steps per task, each a chain of
        // memory round trips (cell, neighbour, operands): what counts is the number of rays in flight per step, and their short
        // queues never overflow - 16 sectors (round 3, config 1: 4 sectors 5.33 ms, 8: 4.15, 16: 3.72; round 2 had measured the
        // opposite with the agent-scope fences and the tree walk of that time). In between, 8 keeps the queues of the first,
        // unmeasured launches short (forcing 16 there: MAGTAR scattering 13.6 against 10.0 s, N-body direct light 27.5 against 20.2 s).
        c->cheap_sources[rt_type & 15] = ips < 4000.0;
        c->G_want[rt_type & 15] = ips < 4000.0 ? 16 : ips < 20000.0 ? 8 : 16;
    }
    return DRT_OK;
}

extern "C" int drt_rt_loop_iobs(drt_ctx* c, int32_t rt_type)
{
    if (!c) return DRT_E_ARG;
    if (!c->part2) return fail(c, DRT_E_ARG, "drt_rt_loop_iobs before drt_rt_prepare(prep_part2)");
    if (rt_type != DRT_RTT_I_OBS_DIR_CELL && rt_type != DRT_RTT_I_OBS_DIR_SRC && rt_type != DRT_RTT_I_OBS)
        return fail(c, DRT_E_ARG, "drt_rt_loop_iobs: rt_type %d is not an i_obs stage", rt_type);
    if (c->ndir == 0) return DRT_OK;  // no observer (rt_routines.f90:1481)
    if (c->par.id_mpi != 0) return DRT_OK;  // no_communications: only the main process (rt_routines.f90:1483)
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    std::vector<int> list;
    if ((rc = build_src_list(c, rt_type, list))) return rc;
    c->cnt.sources = (int64_t)list.size(); c->cnt.kernel_ms = 0; c->cnt.launches = 0;
    if (list.empty()) return DRT_OK;
    long long nrays = (long long)list.size() * c->ndir;
    int R = 64 / c->LP;
    int blocks = (int)((nrays + 4ll * R - 1) / (4ll * R));
    if (blocks > c->ncu * 8) blocks = c->ncu * 8;
    HIPCHK(c, hipEventRecord(c->ev0, 0));
    switch (c->LP) {
    case 1: launch_iobs<1>(c, rt_type, (int)list.size(), blocks); break;
    case 2: launch_iobs<2>(c, rt_type, (int)list.size(), blocks); break;
    case 4: launch_iobs<4>(c, rt_type, (int)list.size(), blocks); break;
    case 8: launch_iobs<8>(c, rt_type, (int)list.size(), blocks); break;
    case 16: launch_iobs<16>(c, rt_type, (int)list.size(), blocks); break;
    case 32: launch_iobs<32>(c, rt_type, (int)list.size(), blocks); break;
    default: launch_iobs<64>(c, rt_type, (int)list.size(), blocks); break;
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev1, 0));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->cnt.kernel_ms = ms; c->cnt.launches = 1;
    if ((rc = check_device_error(c))) return rc;
    return pull_counters(c);
}

// ---- RCCL, loaded at run time so that the library has no link-time dependency on it ------------------
struct drt_uid { char b[DRT_UNIQUE_ID_BYTES]; };  // ncclUniqueId is passed by value: 128 bytes
typedef int (*nccl_get_uid_t)(void*);
typedef int (*nccl_init_rank_t)(void**, int, drt_uid, int);
typedef int (*nccl_allreduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_destroy_t)(void*);
static void* g_rccl = nullptr;
static void* rccl_sym(const char* name)
{
    if (!g_rccl) {
        const char* env = getenv("DRT_RCCL_LIB");
        // a copy the process already holds comes first (a host that has imported torch has torch's librccl.so mapped, built for
        // the HIP runtime the process runs on - see host.py on the one-runtime rule)
        if (!env) for (const char* p : {"librccl.so.1", "librccl.so"}) { g_rccl = dlopen(p, RTLD_NOW | RTLD_NOLOAD); if (g_rccl) break; }
        const char* cand[] = {env, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        if (!g_rccl) for (const char* p : cand) { if (!p) continue; g_rccl = dlopen(p, RTLD_NOW | RTLD_GLOBAL); if (g_rccl) break; }
    }
    return g_rccl ? dlsym(g_rccl, name) : nullptr;
}

extern "C" int drt_comm_unique_id(char id[DRT_UNIQUE_ID_BYTES])
{
    nccl_get_uid_t f = (nccl_get_uid_t)rccl_sym("ncclGetUniqueId");
    if (!f) return DRT_E_COMM;
    return f(id) == 0 ? DRT_OK : DRT_E_COMM;
}

extern "C" int drt_comm_init(drt_ctx* c, int32_t nranks, int32_t rank, const char id[DRT_UNIQUE_ID_BYTES])
{
    if (!c || nranks < 1 || rank < 0 || rank >= nranks) return fail(c, DRT_E_ARG, "drt_comm_init: bad argument");
    if (!c->have_par) return fail(c, DRT_E_ARG, "drt_comm_init: call drt_set_params first");
    if (nranks != c->par.np_mpi || rank != c->par.id_mpi)
        return fail(c, DRT_E_ARG, "drt_comm_init: nranks/rank (%d/%d) differ from np_mpi/id_mpi (%d/%d) of drt_set_params", nranks, rank, c->par.np_mpi, c->par.id_mpi);
    c->nranks = nranks; c->rank = rank;
    if (nranks == 1) return DRT_OK;
    HIPCHK(c, hipSetDevice(c->device));
    nccl_init_rank_t f = (nccl_init_rank_t)rccl_sym("ncclCommInitRank");
    if (!f) return fail(c, DRT_E_COMM, "RCCL library not found (set DRT_RCCL_LIB)");
    drt_uid u;
    memcpy(u.b, id, DRT_UNIQUE_ID_BYTES);
    int r = f(&c->comm, nranks, u, rank);
    if (r != 0) return fail(c, DRT_E_COMM, "ncclCommInitRank failed (%d)", r);
    return DRT_OK;
}

extern "C" int drt_comm_destroy(drt_ctx* c)
{
    if (!c) return DRT_E_ARG;
    if (c->comm) {
        nccl_destroy_t f = (nccl_destroy_t)rccl_sym("ncclCommDestroy");
        if (f) f(c->comm);
        c->comm = nullptr;
    }
    return DRT_OK;
}

// The library's own RCCL path end to end on ONE rank: the library is found and its entry points resolved, a one-rank
// communicator is made from a fresh unique id (passed by value, as ncclCommInitRank wants it), a device buffer goes through
// ncclAllReduce(double, sum) in place on the library's stream and must come back as it went in, the communicator is destroyed.
// What a node with several GPUs adds is the transport; everything this library does with RCCL is exercised here.
extern "C" int drt_comm_selftest(drt_ctx* c)
{
    if (!c) return DRT_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    nccl_get_uid_t f_uid = (nccl_get_uid_t)rccl_sym("ncclGetUniqueId");
    nccl_init_rank_t f_init = (nccl_init_rank_t)rccl_sym("ncclCommInitRank");
    nccl_allreduce_t f_ar = (nccl_allreduce_t)rccl_sym("ncclAllReduce");
    nccl_destroy_t f_des = (nccl_destroy_t)rccl_sym("ncclCommDestroy");
    if (!f_uid || !f_init || !f_ar || !f_des) return fail(c, DRT_E_COMM, "RCCL library not found or incomplete (set DRT_RCCL_LIB)");
    drt_uid u;
    int r = f_uid(u.b);
    if (r != 0) return fail(c, DRT_E_COMM, "ncclGetUniqueId failed (%d)", r);
    void* comm = nullptr;
    if ((r = f_init(&comm, 1, u, 0)) != 0) return fail(c, DRT_E_COMM, "ncclCommInitRank(1 rank) failed (%d)", r);
    const size_t n = 4096;
    std::vector<double> h(n), back(n);
    for (size_t k = 0; k < n; k++) h[k] = 1.0 / (double)(k + 1);
    double* d = nullptr;
    int rc = DRT_OK;
    if (hipMalloc((void**)&d, n * sizeof(double)) != hipSuccess) rc = fail(c, DRT_E_HIP, "hipMalloc");
    if (!rc && hipMemcpy(d, h.data(), n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) rc = fail(c, DRT_E_HIP, "hipMemcpy");
    if (!rc && (r = f_ar(d, d, n, /*ncclFloat64*/ 8, /*ncclSum*/ 0, comm, 0)) != 0) rc = fail(c, DRT_E_COMM, "ncclAllReduce failed (%d)", r);
    if (!rc && hipDeviceSynchronize() != hipSuccess) rc = fail(c, DRT_E_HIP, "hipDeviceSynchronize after ncclAllReduce");
    if (!rc && hipMemcpy(back.data(), d, n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(c, DRT_E_HIP, "hipMemcpy");
    if (!rc) for (size_t k = 0; k < n; k++) if (back[k] != h[k]) { rc = fail(c, DRT_E_COMM, "one-rank all-reduce changed element %zu", k); break; }
    if (d) (void)hipFree(d);
    f_des(comm);
    return rc;
}

static int allreduce_sum(drt_ctx* c, double* d, size_t n)
{
    if (c->par.np_mpi == 1) return DRT_OK;
    nccl_allreduce_t f = (nccl_allreduce_t)rccl_sym("ncclAllReduce");
    if (!f || !c->comm) return fail(c, DRT_E_COMM, "drt_rt_mpi: np_mpi > 1 but no communicator (drt_comm_init)");
    int r = f(d, d, n, /*ncclFloat64*/ 8, /*ncclSum*/ 0, c->comm, 0);
    if (r != 0) return fail(c, DRT_E_COMM, "ncclAllReduce failed (%d)", r);
    return DRT_OK;
}

// rt_mpi (io_routines.f90:3667-3704)
extern "C" int drt_rt_mpi(drt_ctx* c, int32_t rt_type)
{
    if (!c || !c->prepared) return fail(c, DRT_E_ARG, "drt_rt_mpi: not prepared");
    // the sources were shared according to np_mpi/id_mpi (drt_set_params): without a matching communicator the fields
    // of this rank are partial sums and must not be passed on as results
    if (c->par.np_mpi > 1 && (!c->comm || c->nranks != c->par.np_mpi || c->rank != c->par.id_mpi))
        return fail(c, DRT_E_COMM, "drt_rt_mpi: np_mpi = %d but no communicator of that size (drt_comm_init)", c->par.np_mpi);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t NL = (size_t)c->N * c->L;
    int rc;
    if (rt_type == DRT_RTT_PRECALC_SRC) {  // reduce_u_fest_arr (:3708-3729)
        if ((rc = allreduce_sum(c, c->d_u_fest, NL))) return rc;
    } else if (rt_type == DRT_RTT_DIR_SRC || rt_type == DRT_RTT_SCATT) {
        // reduce_u_final_arr (:3732-3761)
        if (rt_type == DRT_RTT_SCATT) hipLaunchKernelGGL(k_u_sub_clamp, dim3(1024), dim3(256), 0, 0, c->d_u_final, c->d_u_fest, (unsigned long long)NL);
        if ((rc = allreduce_sum(c, c->d_u_final, NL))) return rc;
        if (rt_type == DRT_RTT_SCATT) hipLaunchKernelGGL(k_axpy, dim3(1024), dim3(256), 0, 0, c->d_u_final, c->d_u_fest, (unsigned long long)NL);
        // reduce_scaspe_arr (:3806-3834), one wavelength at a time
        for (int il = 0; il < c->L; il++) if ((rc = allreduce_sum(c, c->d_scaspe[il], (size_t)c->N * c->nstride[il]))) return rc;
        // reduce_lum_lost (:3837-3855)
        if (rt_type == DRT_RTT_SCATT && c->par.np_mpi > 1)
            hipLaunchKernelGGL(k_scale_sub, dim3(1), dim3(64), 0, 0, c->d_lum_lost, c->d_lum_lost_prev, (c->par.np_mpi - 1.0) / c->par.np_mpi, c->L);
        if ((rc = allreduce_sum(c, c->d_lum_lost, (size_t)c->L))) return rc;
    } else return fail(c, DRT_E_ARG, "drt_rt_mpi: rt_type %d has no reduction (io_routines.f90:3697)", rt_type);
    // not in the reference: the wave time per cell source of the stage's cell launch, known to the rank that traced it, for
    // every rank's next deal (build_src_list)
    if ((rc = allreduce_sum(c, c->d_src_cost, 2 * (size_t)c->N))) return rc;   // costs and queue lengths
    c->cost_synced = true;
    if (c->psel_av && rt_type != DRT_RTT_PRECALC_SRC) {  // reduce_psel_av_arr (io_routines.f90:3859-3881); sums and counts, a source's tasks may be spread over ranks here
        // every rank issues these two collectives or none does: the condition is the flag alone (drt_rt_loop allocates and
        // zeroes the arrays on every rank, also on one that has no source of its own)
        if (!c->d_psel_sum) {
            if ((rc = dev_zero(c, &c->d_psel_sum, (size_t)c->N + c->nsrc))) return rc;
            if ((rc = dev_zero(c, &c->d_psel_cnt, (size_t)c->N + c->nsrc))) return rc;
        }
        if ((rc = allreduce_sum(c, c->d_psel_sum, (size_t)c->N + c->nsrc))) return rc;
        if ((rc = allreduce_sum(c, c->d_psel_cnt, (size_t)c->N + c->nsrc))) return rc;
    }
    HIPCHK(c, hipDeviceSynchronize());
    return DRT_OK;
}

// print_psel_av (rt_routines.f90:269): keep, for every rt_loop, sum and number of the paths of the rays of each source that ended
extern "C" int drt_set_psel_av(drt_ctx* c, int32_t on)
{
    if (!c) return DRT_E_ARG;
    c->psel_av = on != 0;
    return DRT_OK;
}

extern "C" int drt_set_algorithm(drt_ctx* c, int32_t rta)
{
    if (!c) return DRT_E_ARG;
    if (rta != DRT_RTA_MAIN && rta != DRT_RTA_I_OBS) return fail(c, DRT_E_ARG, "drt_set_algorithm: %d is neither DRT_RTA_MAIN nor DRT_RTA_I_OBS", rta);
    c->algorithm = rta;
    if (c->prepared) return upload_problem(c);
    return DRT_OK;
}

extern "C" int drt_mark_cost_reduced(drt_ctx* c)
{
    if (!c) return DRT_E_ARG;
    c->cost_synced = true;
    return DRT_OK;
}

// ---- module state out ---------------------------------------------------------------------------------
// Device pointer of a field. Host arrays hold `n` elements; the scaspe arrays are kept with padded rows on the device
// (`rows` rows of `cols` values, `stride` apart: DevProblem::nstride), every other field is dense (rows = 0).
static int field_ptr(drt_ctx* c, int which, int il, double** d, size_t* n, size_t* rows = nullptr, size_t* cols = nullptr, size_t* stride = nullptr)
{
    const size_t NL = (size_t)c->N * c->L;
    *d = nullptr; *n = 0;
    if (rows) *rows = 0;
    switch (which) {
    case DRT_A_U_FEST: *d = c->d_u_fest; *n = NL; break;
    case DRT_A_U_FINAL: *d = c->d_u_final; *n = NL; break;
    case DRT_A_LUMCELL: *d = c->d_lumcell; *n = NL; break;
    case DRT_A_DENS: *d = c->d_dens; *n = NL; break;
    case DRT_A_LUM_LOST: *d = c->d_lum_lost; *n = (size_t)c->L; break;
    case DRT_A_LUM_LOST_PREV: *d = c->d_lum_lost_prev; *n = (size_t)c->L; break;
    case DRT_A_SRC_COST: *d = c->d_src_cost; *n = (size_t)c->N; break;
    case DRT_A_PSEL_SUM: *d = c->d_psel_sum; *n = (size_t)c->N + c->nsrc; break;
    case DRT_A_PSEL_CNT: *d = c->d_psel_cnt; *n = (size_t)c->N + c->nsrc; break;
    case DRT_A_I_OBS: *d = c->d_i_obs; *n = (size_t)c->ndir * (c->N + c->nsrc) * c->L; break;
    case DRT_A_SCASPE: case DRT_A_SCASPE_TOT: case DRT_A_SCASPE_PREV: {
        if (il < 0 || il >= c->L) return fail(c, DRT_E_ARG, "wavelength index %d out of range", il);
        auto& v = which == DRT_A_SCASPE ? c->d_scaspe : which == DRT_A_SCASPE_TOT ? c->d_scaspe_tot : c->d_scaspe_prev;
        if ((int)v.size() > il) *d = v[il];
        *n = (size_t)c->N * (c->npix.empty() ? 0 : c->npix[il]);
        if (rows && !c->npix.empty()) { *rows = (size_t)c->N; *cols = (size_t)c->npix[il]; *stride = (size_t)c->nstride[il]; }
        break;
    }
    default: return fail(c, DRT_E_ARG, "unknown field %d", which);
    }
    if (!*d) return fail(c, DRT_E_ARG, "field %d is not allocated at this stage", which);
    return DRT_OK;
}

extern "C" int drt_get_array(drt_ctx* c, int32_t which, int32_t il, double* host, size_t count)
{
    if (!c || !host) return DRT_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (which == DRT_A_TOT_RAD_EN) {
        if (count != (size_t)c->L || c->tot_rad_en.size() != count) return fail(c, DRT_E_ARG, "tot_rad_en: count mismatch");
        memcpy(host, c->tot_rad_en.data(), count * sizeof(double));
        return DRT_OK;
    }
    if (which == DRT_A_PSEL_AV) {  // psel_av of every source of the last rt_loop (rt_routines.f90:1459); 0 where no ray ended
        const size_t M = (size_t)c->N + c->nsrc;
        if (!c->psel_av || !c->d_psel_sum) return fail(c, DRT_E_ARG, "psel_av: not enabled (drt_set_psel_av) or no rt_loop yet");
        if (count != M) return fail(c, DRT_E_ARG, "psel_av: count must be tot_ncell + tot_p_src");
        std::vector<double> s(M), n(M);
        HIPCHK(c, hipMemcpy(s.data(), c->d_psel_sum, M * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(n.data(), c->d_psel_cnt, M * sizeof(double), hipMemcpyDeviceToHost));
        for (size_t k = 0; k < M; k++) host[k] = n[k] > 0 ? s[k] / n[k] : 0.0;
        return DRT_OK;
    }
    if (which == DRT_A_DENS_STARS) {
        if (count != c->dens_stars_h.size()) return fail(c, DRT_E_ARG, "dens_stars_arr: count mismatch");
        memcpy(host, c->dens_stars_h.data(), count * sizeof(double));
        return DRT_OK;
    }
    double* d; size_t n, rows, cols, stride; int rc;
    if ((rc = field_ptr(c, which, il, &d, &n, &rows, &cols, &stride))) return rc;
    if (n != count) return fail(c, DRT_E_ARG, "field %d holds %zu elements, caller passed %zu", which, n, count);
    if (rows) HIPCHK(c, hipMemcpy2D(host, cols * sizeof(double), d, stride * sizeof(double), cols * sizeof(double), rows, hipMemcpyDeviceToHost));
    else HIPCHK(c, hipMemcpy(host, d, n * sizeof(double), hipMemcpyDeviceToHost));
    return DRT_OK;
}

extern "C" int drt_set_array(drt_ctx* c, int32_t which, int32_t il, const double* host, size_t count)
{
    if (!c || !host) return DRT_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t NL = (size_t)c->N * c->L;
    if (which == DRT_A_DENS) {
        if (!c->have_grid || c->L < 1 || count != NL) return fail(c, DRT_E_ARG, "dens_arr: set grid and wavelengths first; count must be lnum*tot_ncell");
        int rc = dev_upload(c, &c->d_dens, host, NL);
        if (rc == DRT_OK && c->prepared) rc = upload_problem(c);
        return rc;
    }
    if (which == DRT_A_DENS_STARS) {
        if (!c->have_grid || c->L < 1 || count != NL) return fail(c, DRT_E_ARG, "dens_stars_arr: set grid and wavelengths first; count must be lnum*tot_ncell");
        c->dens_stars_h.assign(host, host + NL);
        c->prepared = false;
        return DRT_OK;
    }
    double* d; size_t n, rows, cols, stride; int rc;
    if (which == DRT_A_SCASPE_TOT && c->part2 && (c->d_scaspe_tot.empty() || !c->d_scaspe_tot[0])) {
        // create_scaspe_tot (dartray_hub.f90:155, the i_obs algorithm and restarts: scaspe_tot comes from the host)
        c->d_scaspe_tot.assign(c->L, nullptr);
        for (int k = 0; k < c->L; k++) if ((rc = dev_zero(c, &c->d_scaspe_tot[k], (size_t)c->N * c->nstride[k]))) return rc;
        if ((rc = upload_problem(c))) return rc;
    }
    if ((rc = field_ptr(c, which, il, &d, &n, &rows, &cols, &stride))) return rc;
    if (n != count) return fail(c, DRT_E_ARG, "field %d holds %zu elements, caller passed %zu", which, n, count);
    if (rows) HIPCHK(c, hipMemcpy2D(d, stride * sizeof(double), host, cols * sizeof(double), cols * sizeof(double), rows, hipMemcpyHostToDevice));
    else HIPCHK(c, hipMemcpy(d, host, n * sizeof(double), hipMemcpyHostToDevice));
    if (which == DRT_A_LUMCELL) c->lumcell_h.assign(host, host + n);
    return DRT_OK;
}

extern "C" int drt_device_ptr(drt_ctx* c, int32_t which, int32_t il, void** dev_ptr, size_t* count)
{
    if (!c || !dev_ptr || !count) return DRT_E_ARG;
    double* d; size_t n, rows, cols, stride; int rc;
    if ((rc = field_ptr(c, which, il, &d, &n, &rows, &cols, &stride))) return rc;
    *dev_ptr = d; *count = rows ? rows * stride : n;  // scaspe arrays: the padded device array (the padding stays zero)
    if (which == DRT_A_SRC_COST) *count = 2 * n;      // the costs and, behind them, the queue lengths: one sum over the ranks for both
    return DRT_OK;
}

extern "C" int drt_get_npix(const drt_ctx* c, int32_t il, int32_t* npix, int32_t* npix_hp, int32_t* kp)
{
    if (!c || il < 0 || il >= c->L || c->npix.empty()) return DRT_E_ARG;
    if (npix) *npix = c->npix[il];
    if (npix_hp) *npix_hp = c->npix_hp[il];
    if (kp) *kp = c->kp[il];
    return DRT_OK;
}

extern "C" int drt_get_src_cell(const drt_ctx* c, int32_t* src_cell, double* ccoord_p_src)
{
    if (!c || !c->prepared) return DRT_E_ARG;
    if (src_cell) memcpy(src_cell, c->src_cell.data(), (size_t)c->N * sizeof(int32_t));
    if (ccoord_p_src && c->nsrc) memcpy(ccoord_p_src, c->p_src_pos.data(), 3ull * c->nsrc * sizeof(double));
    return DRT_OK;
}

// The cells this rank would trace in drt_rt_loop / drt_rt_loop_iobs(rt_type), in launch order.
extern "C" int drt_get_sources(drt_ctx* c, int32_t rt_type, int32_t* out, int32_t capacity, int32_t* count)
{
    if (!c || !c->prepared || !count) return fail(c, DRT_E_ARG, "drt_get_sources: not prepared");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<int> list;
    int rc = build_src_list(c, rt_type, list);
    if (rc) return rc;
    *count = (int32_t)list.size();
    if (out) for (int i = 0; i < (int)list.size() && i < capacity; i++) out[i] = list[i];
    return DRT_OK;
}

extern "C" double drt_en_lim(const drt_ctx* c) { return c ? c->en_lim : 0.0; }
extern "C" int32_t drt_iterations(const drt_ctx* c) { return c ? c->iterations : -1; }
extern "C" int32_t drt_cnflag(const drt_ctx* c) { return c ? c->cnflag : -1; }

// calc_sed + sum_i_obs (sed_routines.f90:90-146, 211-246), units 'erg/s/Hz/pc^2/sr'; sed(lnum,tot_ndir)
extern "C" int drt_calc_sed(drt_ctx* c, double* sed)
{
    if (!c || !sed || !c->d_i_obs) return fail(c, DRT_E_ARG, "drt_calc_sed: no i_obs array");
    HIPCHK(c, hipSetDevice(c->device));
    const int L = c->L, M = c->N + c->nsrc;
    std::vector<double> io((size_t)c->ndir * M * L);
    HIPCHK(c, hipMemcpy(io.data(), c->d_i_obs, io.size() * sizeof(double), hipMemcpyDeviceToHost));
    const double parsec = (double)3.08567758E16f;
    for (int il = 0; il < L; il++) for (int j = 0; j < c->ndir; j++) {
        double s = 0, s2 = 0;
        for (int i = 0; i < c->N; i++) {
            double v = io[((size_t)j * M + i) * L + il];
            v = v * pow(10.0, -7); v = v * pow(10.0, 26) / pow(parsec, 2.);
            double r = c->csize[i] / c->par.dist_obs;
            s += v * (r * r);
        }
        for (int k = 0; k < c->nsrc; k++) {
            double v = io[((size_t)j * M + c->N + k) * L + il];
            v = v * pow(10.0, -7); v = v * pow(10.0, 26) / pow(parsec, 2.);
            double r = c->csize[c->cell_src[k]] / c->par.dist_obs;
            s2 += v * (r * r);
        }
        sed[(size_t)j * L + il] = s + s2;
    }
    return DRT_OK;
}

// Developer aid (not part of the reference's interface): record every ray launch of the following
// rt_loop calls as (source cell, nside, pixel, cc_old, ray type, start cell).
extern "C" int drt_debug_trace(drt_ctx* c, int capacity, int* out, int* count)
{
    if (!c) return DRT_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (out && c->d_dbg) {
        int n = 0;
        HIPCHK(c, hipMemcpy(&n, c->d_dbg, sizeof(int), hipMemcpyDeviceToHost));
        if (count) *count = n;
        if (n > c->dbg_cap) n = c->dbg_cap;
        HIPCHK(c, hipMemcpy(out, c->d_dbg + 1, (size_t)n * 6 * sizeof(int), hipMemcpyDeviceToHost));
        return DRT_OK;
    }
    if (c->d_dbg) { (void)hipFree(c->d_dbg); c->d_dbg = nullptr; }
    c->dbg_cap = capacity;
    if (capacity > 0) {
        HIPCHK(c, hipMalloc((void**)&c->d_dbg, (1 + 6ull * capacity) * sizeof(int)));
        HIPCHK(c, hipMemset(c->d_dbg, 0, sizeof(int)));
    }
    if (c->prepared) return upload_problem(c);
    return DRT_OK;
}

// Developer aid: change the share of the sources this context traces without preparing it again (tools/partition_ab.py)
extern "C" int drt_debug_set_rank(drt_ctx* c, int np_mpi, int id_mpi)
{
    if (!c || np_mpi < 1 || id_mpi < 0 || id_mpi >= np_mpi) return DRT_E_ARG;
    c->par.np_mpi = np_mpi; c->par.id_mpi = id_mpi;
    return DRT_OK;
}

// Developer aid: DevProblem::dbg_flags (only builds with -DDRT_DEBUG_FLAGS look at them)
extern "C" int drt_debug_flags(drt_ctx* c, int flags)
{
    if (!c) return DRT_E_ARG;
    c->dbg_flags = flags;
    if (c->prepared) return upload_problem(c);
    return DRT_OK;
}

// Developer aid: raw device counters (slots 8.. hold the phase timers of a -DDRT_PROFILE build).
extern "C" int drt_debug_counters(drt_ctx* c, unsigned long long* out, int n)
{
    if (!c || !out || n > CNT_N) return DRT_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy(out, c->d_counters, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return DRT_OK;
}

extern "C" int drt_get_counters(const drt_ctx* c, drt_counters* out)
{
    if (!c || !out) return DRT_E_ARG;
    *out = c->cnt;
    return DRT_OK;
}

extern "C" int drt_reset_counters(drt_ctx* c)
{
    if (!c) return DRT_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemset(c->d_counters, 0, CNT_N * sizeof(unsigned long long)));
    memset(&c->cnt, 0, sizeof c->cnt);
    return DRT_OK;
}
