/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see dartray_oracle.h).
 *
 * CPU restatement of DART-Ray's ray-cell hot path. All file:line citations are into
 * /root/reference. The code keeps the reference's order of floating-point operations, its
 * single-precision literals (promoted to double exactly as Fortran does) and its queue order, so a
 * single-threaded run agrees with the reference to rounding level (tests/test_oracle_golden.py).
 * Build with -ffp-contract=off: the reference is compiled without FMA contraction.
 */
#include "dartray_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- constants with the reference's precision --------------------------------------------- */
/* halfpi = asin(1.) in single precision (smooth_grid_routines.f90:311-317, healpix_routines.f90:13) */
#define F32(x) ((double)(float)(x))
static double HALFPI, PI_, TWOPI, TWOTHIRD;
static const double GLEPSILON = (double)1.0e-7f; /* rt_routines.f90:76 */
static const double TOL_P = (double)1.0e-6f;      /* rt_routines.f90:276 */
static const double TAU_THIN = (double)1.0e-8f;   /* "1E-008" literals, rt_routines.f90:2399-2547 */
static const double LEN_EPS = (double)1.0e-5f;    /* rt_routines.f90:1712 */
static const double OUTCUBE_F = (double)(1.0f + 1.0e-7f); /* (1+1E-7), rt_routines.f90:2079 */
static const double MERGE_TOL = (double)0.001f;   /* ray_list.f90:246 */

enum { RAY_HIGH = 0, RAY_LOW = 1, RAY_RECO = 2, RAY_GONE = 3, RAY_I_OBS = 4 }; /* ray_list.f90:49-59 */
enum { RAS_GO_HIGH = 0, RAS_GONE = 1, RAS_GO_LOW = 2, RAS_FIRST = 3, RAS_RELAUNCH = 4 }; /* rt_routines.f90:255-264 */
#define NSIDE_MIN 4  /* rt_routines.f90:60 */
#define NPIX_MAIN 12 /* rt_routines.f90:62 */

/* ---- HEALPix NESTED (healpix_routines.f90) -------------------------------------------------- */
static int pix2x[1024], pix2y[1024], x2pix1[128], y2pix1[128];
static int tables_ready = 0;

static void mk_tables(void)
{
    /* mk_pix2xy, healpix_routines.f90:145-175 */
    for (int kpix = 0; kpix < 1024; kpix++) {
        int jpix = kpix, ix = 0, iy = 0, ip = 1;
        while (jpix != 0) {
            int id = jpix % 2; jpix /= 2; ix = id * ip + ix;
            id = jpix % 2; jpix /= 2; iy = id * ip + iy;
            ip *= 2;
        }
        pix2x[kpix] = ix; pix2y[kpix] = iy;
    }
    /* mk_xy2pix1, healpix_routines.f90:351-386 */
    for (int i = 0; i < 128; i++) {
        int j = i, k = 0, ip = 1;
        while (j != 0) { int id = j % 2; j /= 2; k = ip * id + k; ip *= 4; }
        x2pix1[i] = k; y2pix1[i] = 2 * k;
    }
    HALFPI = (double)asinf(1.0f);
    PI_ = 2 * HALFPI;
    TWOPI = 2 * PI_;
    TWOTHIRD = (double)(2.0f / 3.0f);
    tables_ready = 1;
}

/* pix2ang_nest, healpix_routines.f90:30-136 (32-bit branch, nside <= 8192) */
void orc_pix2ang_nest(int nside, int ipix, double *theta, double *phi)
{
    static const int jrll[12] = {2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4};
    static const int jpll[12] = {1, 3, 5, 7, 0, 2, 4, 6, 1, 3, 5, 7};
    if (!tables_ready) mk_tables();
    int npface = nside * nside, nl4 = 4 * nside;
    int face_num = ipix / npface;
    int ipf = ipix % npface;
    double fn = (double)nside;
    double fact1 = 1.0 / (3.0 * fn * fn);
    double fact2 = 2.0 / (3.0 * fn);
    int ip_low = ipf & 1023, ip_trunc = ipf / 1024;
    int ip_med = ip_trunc & 1023, ip_hi = ip_trunc / 1024;
    int ix = 1024 * pix2x[ip_hi] + 32 * pix2x[ip_med] + pix2x[ip_low];
    int iy = 1024 * pix2y[ip_hi] + 32 * pix2y[ip_med] + pix2y[ip_low];
    int jrt = ix + iy, jpt = ix - iy;
    int jr = jrll[face_num] * nside - jrt - 1;
    int nr, kshift;
    double z;
    if (jr < nside) { nr = jr; z = 1.0 - nr * fact1 * nr; kshift = 0; }
    else if (jr <= 3 * nside) { nr = nside; z = (2 * nside - jr) * fact2; kshift = (jr - nside) & 1; }
    else { nr = nl4 - jr; z = -1.0 + nr * fact1 * nr; kshift = 0; }
    *theta = acos(z);
    int jp = (jpll[face_num] * nr + jpt + 1 + kshift) / 2;
    if (jp > nl4) jp -= nl4;
    if (jp < 1) jp += nl4;
    *phi = (jp - (kshift + 1) * 0.5) * (HALFPI / nr);
}

/* ang2pix_nest, healpix_routines.f90:238-348 */
int orc_ang2pix_nest(int nside, double theta, double phi)
{
    if (!tables_ready) mk_tables();
    double z = cos(theta), za = fabs(z);
    double m = fmod(phi, TWOPI);
    if (m < 0) m += TWOPI; /* MODULO(phi, twopi) */
    double tt = m / HALFPI;
    int face_num, ix, iy;
    if (za <= TWOTHIRD) {
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
    int ipf = x2pix1[ix & 127] + y2pix1[iy & 127] + (x2pix1[ix / 128] + y2pix1[iy / 128]) * 16384;
    return ipf + face_num * nside * nside;
}

/* ---- ray queues (ray_list.f90:7-33) ---------------------------------------------------------- */
typedef struct {
    double theta, phi, dplane, prev;
    int nside, isel, ray_type, cc_old;
} rayrec;

typedef struct {
    int n, cap, L;
    rayrec *r;
    double *lum; /* [cap][L] */
} rlist;

static void rl_init(rlist *q, int L) { q->n = 0; q->cap = 0; q->L = L; q->r = NULL; q->lum = NULL; }
static void rl_free(rlist *q) { free(q->r); free(q->lum); q->r = NULL; q->lum = NULL; q->n = q->cap = 0; }
static void rl_reserve(rlist *q, int n)
{
    if (n <= q->cap) return;
    int cap = q->cap ? q->cap : 256;
    while (cap < n) cap *= 2;
    q->r = (rayrec *)realloc(q->r, (size_t)cap * sizeof(rayrec));
    q->lum = (double *)realloc(q->lum, (size_t)cap * q->L * sizeof(double));
    q->cap = cap;
}
static void rl_push(rlist *q, const rayrec *r, const double *lum)
{
    rl_reserve(q, q->n + 1);
    q->r[q->n] = *r;
    memcpy(q->lum + (size_t)q->n * q->L, lum, (size_t)q->L * sizeof(double));
    q->n++;
}

/* ---- context --------------------------------------------------------------------------------- */
typedef struct {
    /* thread-private state of the reference (THREADPRIVATE lists rt_routines.f90:527-528, ray_list.f90:61) */
    int *ccindd_nc, *ccindd; /* [max_lvl][3] */
    int clvl_old;
    double pabs_arr[3];
    double src_ccoord[3];
    int nside;
    double *lum_lost_temp, *i_obs_temp;
    unsigned char *iq_a;
    double **ffn_mw;      /* [L][npix_arr] */
    double **ads, **ffn;  /* [nuniq][npix_unique] */
    double **scaspe_temp; /* [L][npix_arr] */
    double *tau_c, *exp_tau_c, *i_av, *en_sca, *u_beam, *ray_intensity, *src_lum, *src_lum_ray, *lum_tmp;
    rlist high, low, ext;
    int no_ray;
    orc_counters cnt;
    int ipsel_av; double psel_av; /* rays of the running source that ended, sum of their paths (rt_routines.f90:71-72, 122-123) */
    int error;
} tstate;

struct orc_ctx {
    int N, max_lvl, base[2], L, nsrc, ndir;
    int kp_sca_max, bm_par, bm_par0, bm_par_sca, bm_par_max;
    int wall_on[3];
    double wall_coord[3][2];
    int max_sca_iterations;
    double modelsize, rad_lim, accuracy, conv_en_lim, dist_obs, cs;
    int pabs_max; /* INTEGER in the reference (rt_routines.f90:22) */
    int64_t basediv[2];
    int basemask[2];
    double baseinv[2];
    int32_t *cchild, *lvl, *src_cell, *cell_src;
    int64_t *cindex;
    double *ccoord, *csize, *dens, *dens_stars;
    double *csize_arr, *carea_arr, *cvol_arr;
    double *ksca_norm, *gsca, *p_src_pos, *p_src_lum, *dir_out;
    int rta_i_obs; /* rt_algorithm_ID == rta_i_obs */
    /* set_npix_arr */
    int *npix_arr, *npix_hp_arr, *kp_sca_arr, *ik_sca_arr;
    int nuniq, *npix_unique, *npix_hp_unique, *kp_unique;
    double **sin_theta_sca, **cos_theta_sca, **sin_phi_sca, **cos_phi_sca;
    /* fields */
    double *u_fest, *u_final, *lumcell, *lum_lost, *lum_lost_prev, *tot_rad_en, *i_obs;
    double **scaspe, **scaspe_tot, **scaspe_prev;
    double en_lim;
    int iterations, cnflag, rt_type;
    int np, id, nthreads;
    orc_counters cnt;
    double *psel_av_last; /* [N + nsrc]: psel_av of every source of the last rt_loop (what rt_routines.f90:1456-1465 stores) */
    int error;
    FILE *trace; /* developer aid: ORC_TRACE=<file> logs every ray launch */
};

#define CC(c, i, k) ((c)->ccoord[(size_t)(i) * 3 + (k)])

/* ---- tree helpers (smooth_grid_routines.f90) -------------------------------------------------- */
/* set_base, smooth_grid_routines.f90:622-649 */
static void set_base(orc_ctx *c)
{
    for (int i = 0; i < 2; i++) {
        c->baseinv[i] = 1.0 / (double)c->base[i];
        int j = 0;
        long base3 = (long)c->base[i] * c->base[i] * c->base[i];
        while (!(base3 < (1L << j))) j++;
        c->basediv[i] = 1L << j;
        c->basemask[i] = (int)(c->basediv[i] - 1);
    }
}

/* cindex_to_ccindd, smooth_grid_routines.f90:793-814. Fills levels 1..clvl only. */
static void cindex_to_ccindd(const orc_ctx *c, int nc, int clvl, int *ccindd)
{
    for (int i = 1; i <= clvl; i++) {
        int ib = (i > 1) ? 1 : 0;
        int64_t j;
        if (i > 1) {
            int64_t div = c->basediv[0];
            for (int k = 0; k < i - 2; k++) div *= c->basediv[1];
            j = ((c->cindex[nc] / div) & c->basemask[1]) - 1;
        } else {
            j = (c->cindex[nc] & c->basemask[0]) - 1;
        }
        int b = c->base[ib];
        ccindd[(i - 1) * 3 + 0] = (int)(j % b);
        ccindd[(i - 1) * 3 + 1] = (int)((j / b) % b);
        ccindd[(i - 1) * 3 + 2] = (int)((j / ((int64_t)b * b)) % b);
    }
}

/* ccindd_to_cc, smooth_grid_routines.f90:820-852 */
static int ccindd_to_cc(const orc_ctx *c, const int *ccindd, int *cc, int *clvl)
{
    int l = 1;
    for (int i = 1; i <= c->max_lvl; i++) {
        int b = c->base[(i > 1) ? 1 : 0];
        int k = (ccindd[(i - 1) * 3 + 2] * b + ccindd[(i - 1) * 3 + 1]) * b + ccindd[(i - 1) * 3 + 0] + 1;
        l = l + k - 1;
        if (c->cchild[l] == -1) { *cc = l; *clvl = i; return 1; }
        l = c->cchild[l];
    }
    return 0;
}

int orc_locate(const orc_ctx *c, const int *ccindd, int *clvl)
{
    int cc = -1, lv = 0;
    if (!ccindd_to_cc(c, ccindd, &cc, &lv)) return -1;
    if (clvl) *clvl = lv;
    return cc;
}
void orc_cindex_to_ccindd(const orc_ctx *c, int cell, int *ccindd) { cindex_to_ccindd(c, cell, c->lvl[cell], ccindd); }

/* increment, smooth_grid_routines.f90:751-769 */
static void increment(const orc_ctx *c, int ilvl, int *val, int inc, int *ovr)
{
    int b = c->base[(ilvl > 1) ? 1 : 0];
    *val = *val + inc;
    *ovr = inc;
    if (*val >= b) *val = 0;
    else if (*val < 0) *val = b - 1;
    else *ovr = 0;
}

/* ---- set-up ------------------------------------------------------------------------------------- */
/* set_npix_arr, rt_routines.f90:5235-5315 */
static void set_npix_arr(orc_ctx *c)
{
    int L = c->L;
    c->npix_arr = (int *)calloc(L, sizeof(int));
    c->npix_hp_arr = (int *)calloc(L, sizeof(int));
    c->kp_sca_arr = (int *)calloc(L, sizeof(int));
    c->ik_sca_arr = (int *)calloc(L, sizeof(int));
    const int nmin = 5;
    const double g_th = 2 * F32(1e-3);
    for (int i = 0; i < L; i++) {
        double g = fabs(c->gsca[i]);
        if (g < g_th) { c->npix_hp_arr[i] = 1; c->kp_sca_arr[i] = -1; continue; }
        double pmax = (1 - g * g) / pow(1 + g * g - 2 * g, 1.5);
        double pmin = (1 - g * g) / pow(1 + g * g + 2 * g, 1.5);
        double cos_hm = (pow(2 * (1 - g * g) / (pmax + pmin), (double)(2.0f / 3.0f)) - (1 + g * g)) / (-2 * g);
        double fwhm = 2 * acos(cos_hm);
        double delta_min = fwhm / nmin;
        int k_min = (int)(1 / (2 * (double)log10f(2.0f)) * log10(4 * PI_ / (12 * delta_min * delta_min)));
        if (k_min > c->kp_sca_max) k_min = c->kp_sca_max;
        c->kp_sca_arr[i] = k_min;
        c->npix_hp_arr[i] = 12 * (1 << (2 * k_min));
    }
    for (int i = 0; i < L; i++) c->npix_arr[i] = c->npix_hp_arr[i] + c->ndir; /* tot_ndir_scaspe = tot_ndir */
    c->nuniq = 0;
    c->npix_unique = (int *)calloc(c->kp_sca_max + 2, sizeof(int));
    c->npix_hp_unique = (int *)calloc(c->kp_sca_max + 2, sizeof(int));
    c->kp_unique = (int *)calloc(c->kp_sca_max + 2, sizeof(int));
    for (int i = 0; i < L; i++) c->ik_sca_arr[i] = -1;
    for (int k = 0; k <= c->kp_sca_max; k++) {
        int inc = 0;
        for (int i = 0; i < L; i++) if (c->npix_hp_arr[i] == 12 * (1 << (2 * k))) inc = 1;
        if (!inc) continue;
        int j = c->nuniq++;
        c->kp_unique[j] = k;
        c->npix_hp_unique[j] = 12 * (1 << (2 * k));
        c->npix_unique[j] = c->npix_hp_unique[j] + c->ndir;
        for (int i = 0; i < L; i++) if (c->kp_sca_arr[i] == k) c->ik_sca_arr[i] = j;
    }
}

/* create_scaspe, rt_routines.f90:2784-2852: direction tables of the scaspe slots */
static void create_scaspe_dirs(orc_ctx *c)
{
    c->sin_theta_sca = (double **)calloc(c->nuniq + 1, sizeof(double *));
    c->cos_theta_sca = (double **)calloc(c->nuniq + 1, sizeof(double *));
    c->sin_phi_sca = (double **)calloc(c->nuniq + 1, sizeof(double *));
    c->cos_phi_sca = (double **)calloc(c->nuniq + 1, sizeof(double *));
    for (int i = 0; i < c->nuniq; i++) {
        int n = c->npix_unique[i], nhp = c->npix_hp_unique[i];
        int nside_sca = 1 << c->kp_unique[i];
        c->sin_theta_sca[i] = (double *)malloc(n * sizeof(double));
        c->cos_theta_sca[i] = (double *)malloc(n * sizeof(double));
        c->sin_phi_sca[i] = (double *)malloc(n * sizeof(double));
        c->cos_phi_sca[i] = (double *)malloc(n * sizeof(double));
        for (int j = 0; j < n; j++) {
            double th, ph;
            if (j < nhp) orc_pix2ang_nest(nside_sca, j, &th, &ph);
            else { th = c->dir_out[j - nhp]; ph = c->dir_out[c->ndir + (j - nhp)]; }
            c->sin_theta_sca[i][j] = sin(th); c->cos_theta_sca[i][j] = cos(th);
            c->sin_phi_sca[i][j] = sin(ph); c->cos_phi_sca[i][j] = cos(ph);
        }
    }
}

/* prepare_p_src + fix_ccoord_p_src, rt_routines.f90:535-601 */
static void prepare_p_src(orc_ctx *c)
{
    for (int i = 0; i < c->nsrc; i++) {
        int found = 0;
        for (int j = 0; j < c->N && !found; j++) {
            if (c->cchild[j] != -1) continue;
            double rel[3], cellsize = c->csize[j];
            int in = 0;
            for (int k = 0; k < 3; k++) {
                rel[k] = c->p_src_pos[i * 3 + k] - CC(c, j, k);
                if (fabs(rel[k]) < (1. + TOL_P) * cellsize / 2.) in++;
            }
            if (in == 3) {
                c->src_cell[j] = i; c->cell_src[i] = j;
                for (int k = 0; k < 3; k++) {
                    if (fabs(rel[k]) >= cellsize / 2.) {
                        if (rel[k] > 0) c->p_src_pos[i * 3 + k] -= TOL_P * cellsize;
                        else c->p_src_pos[i * 3 + k] += TOL_P * cellsize;
                    }
                }
                found = 1;
            }
        }
        if (!found) { fprintf(stderr, "oracle: host cell of point source %d not found\n", i); c->error = 1; }
    }
}

/* x**n for integer n >= 0 the way a Fortran runtime does it (binary exponentiation) */
static double ipow(double x, int n)
{
    double r = 1.0;
    while (n > 0) { if (n & 1) r *= x; x *= x; n >>= 1; }
    return r;
}

static void finish_setup(orc_ctx *c)
{
    if (!tables_ready) mk_tables();
    set_base(c);
    /* make_csize_arr, io_routines.f90:669-696 */
    c->csize_arr = (double *)malloc((c->max_lvl + 1) * sizeof(double));
    c->carea_arr = (double *)malloc((c->max_lvl + 1) * sizeof(double));
    c->cvol_arr = (double *)malloc((c->max_lvl + 1) * sizeof(double));
    c->pabs_max = (int)(c->modelsize * 10);
    for (int i = 0; i <= c->max_lvl; i++) {
        double cellsize = c->modelsize;
        if (i > 0) cellsize = c->modelsize * (c->baseinv[0] * ipow(c->baseinv[1], i - 1));
        c->csize_arr[i] = cellsize;
        c->carea_arr[i] = cellsize * cellsize;
        c->cvol_arr[i] = cellsize * cellsize * cellsize;
    }
    set_npix_arr(c);
    create_scaspe_dirs(c);
    size_t NL = (size_t)c->N * c->L;
    c->u_fest = (double *)calloc(NL, sizeof(double));
    c->u_final = (double *)calloc(NL, sizeof(double));
    c->lumcell = (double *)calloc(NL, sizeof(double));
    c->lum_lost = (double *)calloc(c->L, sizeof(double));
    c->lum_lost_prev = (double *)calloc(c->L, sizeof(double));
    c->tot_rad_en = (double *)calloc(c->L, sizeof(double));
    c->i_obs = (double *)calloc((size_t)(c->ndir > 0 ? c->ndir : 1) * (c->N + c->nsrc) * c->L, sizeof(double));
    c->psel_av_last = (double *)calloc((size_t)(c->N + c->nsrc), sizeof(double));
    c->scaspe = (double **)calloc(c->L, sizeof(double *));
    c->scaspe_tot = (double **)calloc(c->L, sizeof(double *));
    c->scaspe_prev = (double **)calloc(c->L, sizeof(double *));
    for (int il = 0; il < c->L; il++) c->scaspe[il] = (double *)calloc((size_t)c->N * c->npix_arr[il], sizeof(double));
    c->src_cell = (int32_t *)malloc(c->N * sizeof(int32_t));
    for (int i = 0; i < c->N; i++) c->src_cell[i] = -1;
    c->cell_src = (int32_t *)calloc(c->nsrc > 0 ? c->nsrc : 1, sizeof(int32_t));
    /* rt_prepare, case rtt_start (dartray_hub.f90:390-395) */
    /* set_en_lim, rt_routines.f90:3137-3158 */
    int leaves = 0;
    for (int i = 0; i < c->N; i++) leaves += (c->cchild[i] == -1);
    int tot_sources = leaves + c->nsrc;
    c->en_lim = c->accuracy / (tot_sources * 0.25);
    prepare_p_src(c);
    /* calc_total_luminosity, rt_routines.f90:605-660 */
    for (int i = 0; i < c->N; i++) {
        if (c->cchild[i] != -1) continue;
        double v = c->csize[i] * c->csize[i] * c->csize[i];
        for (int il = 0; il < c->L; il++) c->lumcell[(size_t)i * c->L + il] = c->dens_stars[(size_t)i * c->L + il] * v;
    }
    c->iterations = 0; c->cnflag = 0;
    c->np = 1; c->id = 0; c->nthreads = 1;
    c->trace = getenv("ORC_TRACE") ? fopen(getenv("ORC_TRACE"), "w") : NULL;
    c->bm_par0 = c->bm_par;
}

static void *dupmem(const void *p, size_t n) { void *q = malloc(n ? n : 1); if (p && n) memcpy(q, p, n); return q; }

orc_ctx *orc_create(int ncell, int max_lvl, const int *base, int lnum, int nsrc, int ndir,
                    int kp_sca_max, int bm_par, int bm_par_sca, int bm_par_max,
                    const int *wall_on, const double *wall_coord, int max_sca_iterations,
                    double modelsize, double rad_lim, double accuracy, double conv_en_lim,
                    double dist_obs, double cs,
                    const int32_t *cchild, const int32_t *lvl, const int64_t *cindex,
                    const double *ccoord, const double *csize, const double *dens,
                    const double *dens_stars, const double *ksca_norm, const double *gsca,
                    const double *p_src_pos, const double *p_src_lum, const double *dir_out)
{
    orc_ctx *c = (orc_ctx *)calloc(1, sizeof(orc_ctx));
    c->N = ncell; c->max_lvl = max_lvl; c->base[0] = base[0]; c->base[1] = base[1];
    c->L = lnum; c->nsrc = nsrc; c->ndir = ndir; c->kp_sca_max = kp_sca_max;
    c->bm_par = bm_par; c->bm_par_sca = bm_par_sca; c->bm_par_max = bm_par_max;
    for (int k = 0; k < 3; k++) {
        c->wall_on[k] = wall_on ? wall_on[k] : 0;
        c->wall_coord[k][0] = wall_coord ? wall_coord[2 * k] : 0;
        c->wall_coord[k][1] = wall_coord ? wall_coord[2 * k + 1] : 0;
    }
    c->max_sca_iterations = max_sca_iterations;
    c->modelsize = modelsize; c->rad_lim = rad_lim; c->accuracy = accuracy;
    c->conv_en_lim = conv_en_lim; c->dist_obs = dist_obs; c->cs = cs;
    size_t N = ncell, L = lnum;
    c->cchild = (int32_t *)dupmem(cchild, N * 4);
    c->lvl = (int32_t *)dupmem(lvl, N * 4);
    c->cindex = (int64_t *)dupmem(cindex, N * 8);
    c->ccoord = (double *)dupmem(ccoord, N * 24);
    c->csize = (double *)dupmem(csize, N * 8);
    c->dens = (double *)dupmem(dens, N * L * 8);
    c->dens_stars = (double *)dupmem(dens_stars, N * L * 8);
    c->ksca_norm = (double *)dupmem(ksca_norm, L * 8);
    c->gsca = (double *)dupmem(gsca, L * 8);
    c->p_src_pos = (double *)dupmem(p_src_pos, (size_t)nsrc * 24);
    c->p_src_lum = (double *)dupmem(p_src_lum, (size_t)nsrc * L * 8);
    c->dir_out = (double *)dupmem(dir_out, (size_t)ndir * 16);
    finish_setup(c);
    return c;
}

orc_ctx *orc_create_from_file(const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    int32_t hdr[16];
    double dbl[12];
    if (fread(hdr, 4, 16, f) != 16 || fread(dbl, 8, 12, f) != 12 || hdr[0] != 0x44525431) { fclose(f); return NULL; }
    size_t N = hdr[1], L = hdr[5], nsrc = hdr[6], ndir = hdr[7];
    int32_t *cchild = (int32_t *)malloc(N * 4), *lvl = (int32_t *)malloc(N * 4);
    int64_t *cindex = (int64_t *)malloc(N * 8);
    double *ccoord = (double *)malloc(N * 24), *csize = (double *)malloc(N * 8);
    double *dens = (double *)malloc(N * L * 8), *dstar = (double *)malloc(N * L * 8);
    double *ksca = (double *)malloc(L * 8), *gsca = (double *)malloc(L * 8);
    double *ppos = (double *)malloc(nsrc * 24 + 8), *plum = (double *)malloc(nsrc * L * 8 + 8), *dirs = (double *)malloc(ndir * 16 + 8);
    size_t ok = 1;
    ok &= fread(cchild, 4, N, f) == N; ok &= fread(lvl, 4, N, f) == N; ok &= fread(cindex, 8, N, f) == N;
    ok &= fread(ccoord, 8, 3 * N, f) == 3 * N; ok &= fread(csize, 8, N, f) == N;
    ok &= fread(dens, 8, N * L, f) == N * L; ok &= fread(dstar, 8, N * L, f) == N * L;
    ok &= fread(ksca, 8, L, f) == L; ok &= fread(gsca, 8, L, f) == L;
    if (nsrc) { ok &= fread(ppos, 8, 3 * nsrc, f) == 3 * nsrc; ok &= fread(plum, 8, nsrc * L, f) == nsrc * L; }
    if (ndir) ok &= fread(dirs, 8, 2 * ndir, f) == 2 * ndir;
    fclose(f);
    orc_ctx *c = NULL;
    if (ok) {
        int base[2] = {hdr[3], hdr[4]};
        int wall_on[3] = {hdr[12], hdr[13], hdr[14]};
        c = orc_create((int)N, hdr[2], base, (int)L, (int)nsrc, (int)ndir, hdr[8], hdr[9], hdr[10], hdr[11],
                       wall_on, dbl + 6, hdr[15], dbl[0], dbl[1], dbl[2], dbl[3], dbl[4], dbl[5],
                       cchild, lvl, cindex, ccoord, csize, dens, dstar, ksca, gsca, ppos, plum, dirs);
    }
    free(cchild); free(lvl); free(cindex); free(ccoord); free(csize); free(dens); free(dstar);
    free(ksca); free(gsca); free(ppos); free(plum); free(dirs);
    return c;
}

static void free2(double **p, int n) { if (!p) return; for (int i = 0; i < n; i++) free(p[i]); free(p); }

void orc_destroy(orc_ctx *c)
{
    if (!c) return;
    free(c->cchild); free(c->lvl); free(c->src_cell); free(c->cell_src); free(c->cindex);
    free(c->ccoord); free(c->csize); free(c->dens); free(c->dens_stars);
    free(c->csize_arr); free(c->carea_arr); free(c->cvol_arr);
    free(c->ksca_norm); free(c->gsca); free(c->p_src_pos); free(c->p_src_lum); free(c->dir_out);
    free(c->npix_arr); free(c->npix_hp_arr); free(c->kp_sca_arr); free(c->ik_sca_arr);
    free(c->npix_unique); free(c->npix_hp_unique); free(c->kp_unique);
    free2(c->sin_theta_sca, c->nuniq); free2(c->cos_theta_sca, c->nuniq);
    free2(c->sin_phi_sca, c->nuniq); free2(c->cos_phi_sca, c->nuniq);
    free(c->u_fest); free(c->u_final); free(c->lumcell); free(c->lum_lost); free(c->lum_lost_prev);
    free(c->tot_rad_en); free(c->i_obs); free(c->psel_av_last);
    free2(c->scaspe, c->L); free2(c->scaspe_tot, c->L); free2(c->scaspe_prev, c->L);
    if (c->trace) fclose(c->trace);
    free(c);
}

void orc_set_parallel(orc_ctx *c, int np, int id, int nthreads) { c->np = np; c->id = id; c->nthreads = nthreads < 1 ? 1 : nthreads; }
/* rt_algorithm_ID (rt_routines.f90:232-250): 0 = rta_main, 1 = rta_i_obs - the only place where the path looks at it is the
 * source luminosity of rt_loop_iobs(rtt_i_obs), rt_routines.f90:1538-1548 */
/* psel_av of every source (cell index, or N + point-source index) of the last rt_loop; 0 where no ray ended */
const double *orc_psel_av(const orc_ctx *c) { return c->psel_av_last; }
void orc_set_algorithm(orc_ctx *c, int rta) { c->rta_i_obs = (rta == 1); }
/* new observer directions (dartray_i_obs reads another file_dir_out than the main run): theta[ndir] then phi[ndir]; i_obs is zeroed
 * like create_i_obs does (rt_routines.f90:2970-2980) */
void orc_set_dir_out(orc_ctx *c, const double *dir_out)
{
    memcpy(c->dir_out, dir_out, (size_t)c->ndir * 16);
    memset(c->i_obs, 0, (size_t)c->ndir * (c->N + c->nsrc) * c->L * sizeof(double));
}
/* calc_total_luminosity, rt_routines.f90:605-660: lumcell back to the stellar emission (dartray_i_obs, dartray_hub.f90:139) */
void orc_calc_total_luminosity(orc_ctx *c)
{
    for (int i = 0; i < c->N; i++) {
        for (int il = 0; il < c->L; il++) c->lumcell[(size_t)i * c->L + il] = 0.0;
        if (c->cchild[i] != -1) continue;
        double v = c->csize[i] * c->csize[i] * c->csize[i];
        for (int il = 0; il < c->L; il++) c->lumcell[(size_t)i * c->L + il] = c->dens_stars[(size_t)i * c->L + il] * v;
    }
}

/* ---- thread state ------------------------------------------------------------------------------- */
static void ts_init(const orc_ctx *c, tstate *t)
{
    memset(t, 0, sizeof(*t));
    int L = c->L;
    t->ccindd_nc = (int *)calloc(c->max_lvl * 3, sizeof(int));
    t->ccindd = (int *)calloc(c->max_lvl * 3, sizeof(int));
    t->lum_lost_temp = (double *)calloc(L, sizeof(double));
    t->i_obs_temp = (double *)calloc(L, sizeof(double));
    t->iq_a = (unsigned char *)calloc(L, 1);
    t->ffn_mw = (double **)calloc(L, sizeof(double *));
    t->scaspe_temp = (double **)calloc(L, sizeof(double *));
    for (int il = 0; il < L; il++) {
        t->ffn_mw[il] = (double *)calloc(c->npix_arr[il], sizeof(double));
        t->scaspe_temp[il] = (double *)calloc(c->npix_arr[il], sizeof(double));
    }
    t->ads = (double **)calloc(c->nuniq + 1, sizeof(double *));
    t->ffn = (double **)calloc(c->nuniq + 1, sizeof(double *));
    for (int i = 0; i < c->nuniq; i++) {
        t->ads[i] = (double *)calloc(c->npix_unique[i], sizeof(double));
        t->ffn[i] = (double *)calloc(c->npix_unique[i], sizeof(double));
    }
    double **arrs[] = {&t->tau_c, &t->exp_tau_c, &t->i_av, &t->en_sca, &t->u_beam, &t->ray_intensity,
                       &t->src_lum, &t->src_lum_ray, &t->lum_tmp};
    for (size_t k = 0; k < sizeof(arrs) / sizeof(arrs[0]); k++) *arrs[k] = (double *)calloc(L, sizeof(double));
    rl_init(&t->high, L); rl_init(&t->low, L); rl_init(&t->ext, L);
}

static void ts_free(const orc_ctx *c, tstate *t)
{
    free(t->ccindd_nc); free(t->ccindd); free(t->lum_lost_temp); free(t->i_obs_temp); free(t->iq_a);
    free2(t->ffn_mw, c->L); free2(t->scaspe_temp, c->L); free2(t->ads, c->nuniq); free2(t->ffn, c->nuniq);
    free(t->tau_c); free(t->exp_tau_c); free(t->i_av); free(t->en_sca); free(t->u_beam);
    free(t->ray_intensity); free(t->src_lum); free(t->src_lum_ray); free(t->lum_tmp);
    rl_free(&t->high); rl_free(&t->low); rl_free(&t->ext);
}

static inline void atomic_add(double *p, double v)
{
#ifdef _OPENMP
#pragma omp atomic
#endif
    *p += v;
}

/* ---- ray queues ------------------------------------------------------------------------------- */
/* add_to_high_ray_list, ray_list.f90:463-521 */
static void add_to_high(tstate *t, double theta, double phi, int nside, double prev, int isel_old,
                        double dplane, const double *src_lum, int cc, int ray_type)
{
    rayrec r;
    r.theta = theta; r.phi = phi; r.nside = nside;
    if (isel_old != 0) { r.dplane = dplane; r.prev = prev; }
    else { r.dplane = 0.; r.prev = 0.; }
    r.isel = isel_old; r.ray_type = ray_type; r.cc_old = cc;
    rl_push(&t->high, &r, src_lum);
    if (t->high.n > t->cnt.max_high) t->cnt.max_high = t->high.n;
}

/* add_to_low_ray_list, ray_list.f90:532-579 (cproj here is already 1/cproj) */
static void add_to_low(tstate *t, double theta, double phi, int nside, double prev, int isel_old,
                       const double *cproj, const double *src_lum, int cc)
{
    rayrec r;
    r.theta = theta; r.phi = phi; r.nside = nside;
    r.dplane = prev / cproj[isel_old - 1];
    r.prev = prev; r.isel = isel_old; r.ray_type = RAY_LOW; r.cc_old = cc;
    rl_push(&t->low, &r, src_lum);
    if (t->low.n > t->cnt.max_low) t->cnt.max_low = t->low.n;
}

/* create_high_ray_list, ray_list.f90:71-106 */
static void create_high_ray_list(tstate *t, int idir, int idir2, const double *src_lum, int src_id, int nside_low)
{
    t->high.n = 0;
    int ipix = (int)(idir * nside_low * nside_low + idir2 * nside_low * nside_low / 4.);
    rayrec r;
    orc_pix2ang_nest(nside_low, ipix, &r.theta, &r.phi);
    r.nside = nside_low; r.dplane = 0; r.prev = 0; r.isel = 0; r.ray_type = RAY_HIGH; r.cc_old = src_id;
    rl_push(&t->high, &r, src_lum);
}

/* extract_ray_list, ray_list.f90:110-306 (+ compress_high_list:361, compress_low_list:406) */
static void extract_ray_list(tstate *t, int nside)
{
    int L = t->ext.L;
    t->no_ray = 0;
    t->ext.n = 0;
    if (t->high.n > 0) {
        int ni = 0;
        for (int i = 0; i < t->high.n; i++) if (t->high.r[i].nside == nside / 2) ni++;
        if (ni == 0) { t->no_ray = 1; return; }
        int w = 0;
        for (int i = 0; i < t->high.n; i++) {
            if (t->high.r[i].nside == nside / 2) rl_push(&t->ext, &t->high.r[i], t->high.lum + (size_t)i * L);
            else {
                if (w != i) { t->high.r[w] = t->high.r[i]; memmove(t->high.lum + (size_t)w * L, t->high.lum + (size_t)i * L, L * sizeof(double)); }
                w++;
            }
        }
        t->high.n = w;
    } else if (t->low.n > 0) {
        int ni = 0;
        for (int i = 0; i < t->low.n; i++) if (t->low.r[i].nside == nside * 2) ni++;
        if (ni == 0) { t->no_ray = 1; return; }
        int w = 0;
        for (int i = 0; i < t->low.n; i++) {
            if (t->low.r[i].nside == nside * 2) rl_push(&t->ext, &t->low.r[i], t->low.lum + (size_t)i * L);
            else {
                if (w != i) { t->low.r[w] = t->low.r[i]; memmove(t->low.lum + (size_t)w * L, t->low.lum + (size_t)i * L, L * sizeof(double)); }
                w++;
            }
        }
        t->low.n = w;
        /* ray_list.f90:211-296: group four siblings falling in the same coarser pixel */
        int *ipix_arr = (int *)malloc(ni * sizeof(int));
        for (int i = 0; i < ni; i++) ipix_arr[i] = orc_ang2pix_nest(nside, t->ext.r[i].theta, t->ext.r[i].phi);
        if (getenv("ORC_DEBUG_MERGE")) {
            for (int i = 0; i < ni; i++) for (int j = i + 1; j < ni; j++) {
                if (ipix_arr[i] != ipix_arr[j]) continue;
                int pi_ = orc_ang2pix_nest(2 * nside, t->ext.r[i].theta, t->ext.r[i].phi), pj = orc_ang2pix_nest(2 * nside, t->ext.r[j].theta, t->ext.r[j].phi);
                if (pi_ == pj) fprintf(stderr, "oracle: two extracted rays share pixel %d at nside %d: i=%d (type %d isel %d dpl %.17g prev %.17g cc %d) j=%d (type %d isel %d dpl %.17g prev %.17g cc %d)\n",
                    pi_, 2 * nside, i, t->ext.r[i].ray_type, t->ext.r[i].isel, t->ext.r[i].dplane, t->ext.r[i].prev, t->ext.r[i].cc_old,
                    j, t->ext.r[j].ray_type, t->ext.r[j].isel, t->ext.r[j].dplane, t->ext.r[j].prev, t->ext.r[j].cc_old);
            }
        }
        for (int i = 0; i < ni; i++) {
            int ic_arr[5] = {-1, -1, -1, -1, -1};
            if (t->ext.r[i].ray_type == RAY_GONE) continue;
            double dplane_ray = t->ext.r[i].dplane;
            int isel_ray = t->ext.r[i].isel;
            int ipix = ipix_arr[i];
            double theta, phi;
            orc_pix2ang_nest(nside, ipix, &theta, &phi);
            int icount = 1;
            ic_arr[icount] = i;
            for (int j = i + 1; j < ni; j++) {
                double dplane_add = t->ext.r[j].dplane;
                if (ipix_arr[j] == ipix && isel_ray == t->ext.r[j].isel &&
                    (fabs(dplane_ray - dplane_add) / fabs(dplane_ray) < MERGE_TOL)) {
                    icount++;
                    ic_arr[icount] = j;
                    if (t->ext.r[j].ray_type == RAY_GONE && getenv("ORC_DEBUG_MERGE")) {
                        fprintf(stderr, "oracle: entry j=%d already gone is collected again by i=%d (nside %d parent pix %d)\n", j, i, nside, ipix);
                        for (int q = 0; q < ni; q++) if (ipix_arr[q] == ipix)
                            fprintf(stderr, "   q=%d type %d isel %d dplane %.17g prev %.17g cc_old %d theta %.17g phi %.17g own pix %d\n", q, t->ext.r[q].ray_type,
                                    t->ext.r[q].isel, t->ext.r[q].dplane, t->ext.r[q].prev, t->ext.r[q].cc_old, t->ext.r[q].theta, t->ext.r[q].phi,
                                    orc_ang2pix_nest(2 * nside, t->ext.r[q].theta, t->ext.r[q].phi));
                    }
                }
                if (icount == 4) break;
            }
            if (icount == 4) {
                t->ext.r[i].nside = nside;
                t->ext.r[i].theta = theta;
                t->ext.r[i].phi = phi;
                double *li = t->ext.lum + (size_t)i * L;
                for (int iw = 0; iw < L; iw++) {
                    for (int ic = 2; ic <= 4; ic++) li[iw] = li[iw] + t->ext.lum[(size_t)ic_arr[ic] * L + iw];
                    li[iw] = li[iw] / 4;
                }
                /* sum(ext_ray_list(ic_arr)%prev)/4 */
                double s = t->ext.r[ic_arr[1]].prev + t->ext.r[ic_arr[2]].prev + t->ext.r[ic_arr[3]].prev + t->ext.r[ic_arr[4]].prev;
                t->ext.r[i].prev = s / 4;
                for (int ic = 2; ic <= 4; ic++) t->ext.r[ic_arr[ic]].ray_type = RAY_GONE;
            } else {
                for (int j = 1; j <= icount; j++) {
                    rayrec *e = &t->ext.r[ic_arr[j]];
                    add_to_high(t, e->theta, e->phi, nside, e->prev, e->isel, e->dplane,
                                t->ext.lum + (size_t)ic_arr[j] * L, e->cc_old, RAY_RECO);
                    e->ray_type = RAY_GONE;
                }
            }
        }
        free(ipix_arr);
    } else {
        fprintf(stderr, "oracle: extract_ray_list with empty queues\n");
        t->error = 1;
        t->no_ray = 1;
    }
    if (t->ext.n > t->cnt.max_ext) t->cnt.max_ext = t->ext.n;
}

/* define_next_level, rt_routines.f90:2745-2770 */
static int define_next_level(tstate *t)
{
    if (t->high.n > 0) { t->nside = t->nside * 2; return 0; }
    if (t->low.n > 0) { t->nside = t->nside / 2; return 0; }
    return 1;
}

/* ---- geometry ------------------------------------------------------------------------------------ */
/* calc_psel, rt_routines.f90:2307-2351 (cproj is 1/cproj here) */
static void calc_psel(const orc_ctx *c, tstate *t, const int *incvec, double *psel, int *isel, int clvl, int cc, const double *cproj)
{
    double pabs[3];
    for (int k = 0; k < 3; k++) pabs[k] = (double)c->pabs_max;
    if (clvl == t->clvl_old) {
        for (int k = 0; k < 3; k++) if (incvec[k] != 0) pabs[k] = cproj[k] * CC(c, cc, k) + t->pabs_arr[k];
    } else {
        for (int k = 0; k < 3; k++) {
            if (incvec[k] != 0) pabs[k] = cproj[k] * (CC(c, cc, k) - t->src_ccoord[k] + 0.5 * (double)incvec[k] * c->csize_arr[clvl]);
            t->pabs_arr[k] = cproj[k] * ((-t->src_ccoord[k]) + 0.5 * (double)incvec[k] * c->csize_arr[clvl]);
        }
    }
    if (pabs[0] < pabs[2]) {
        if (pabs[0] < pabs[1]) { *psel = pabs[0]; *isel = 1; }
        else { *psel = pabs[1]; *isel = 2; }
    } else {
        if (pabs[2] < pabs[1]) { *psel = pabs[2]; *isel = 3; }
        else { *psel = pabs[1]; *isel = 2; }
    }
}

/* find_cc_new2 + calc_outcube, rt_routines.f90:1962-2085 (cproj is the direction here) */
static int find_cc_new2(const orc_ctx *c, tstate *t, int cc_old, int *cc_new, const double *cproj, double prev_ray, int *out, int *clvl)
{
    double pcoord[3], rel[3];
    *out = 0;
    for (int k = 0; k < 3; k++) pcoord[k] = t->src_ccoord[k] + prev_ray * cproj[k];
    for (int k = 0; k < 3; k++) if (fabs(pcoord[k]) > c->modelsize / 2.) { *out = 1; return 1; }
    double cellsize = c->csize[cc_old];
    int sum = 0;
    for (int k = 0; k < 3; k++) { rel[k] = pcoord[k] - CC(c, cc_old, k); if (fabs(rel[k]) > OUTCUBE_F * cellsize / 2.0) sum++; }
    if (sum == 0) {
        *clvl = c->lvl[cc_old];
        *cc_new = cc_old;
        cindex_to_ccindd(c, *cc_new, *clvl, t->ccindd);
        return 1;
    }
    double np_[3];
    for (int k = 0; k < 3; k++) np_[k] = pcoord[k] + c->modelsize / 2.;
    for (int i = 1; i <= c->max_lvl; i++) {
        int b = c->base[(i > 1) ? 1 : 0];
        for (int k = 0; k < 3; k++) {
            int v = (int)(np_[k] / c->csize_arr[i]);
            if (v == b) v = b - 1;
            t->ccindd[(i - 1) * 3 + k] = v;
        }
        for (int k = 0; k < 3; k++) np_[k] = np_[k] - t->ccindd[(i - 1) * 3 + k] * c->csize_arr[i];
    }
    if (!ccindd_to_cc(c, t->ccindd, cc_new, clvl)) { fprintf(stderr, "oracle: cell not found in find_cc_new2\n"); t->error = 1; return 0; }
    cellsize = c->csize_arr[*clvl];
    sum = 0;
    for (int k = 0; k < 3; k++) { rel[k] = pcoord[k] - CC(c, *cc_new, k); if (fabs(rel[k]) > OUTCUBE_F * cellsize / 2.0) sum++; }
    if (sum != 0) { fprintf(stderr, "oracle: host cell not found (find_cc_new2)\n"); t->error = 1; return 0; }
    return 1;
}

/* ---- scattering phase function ------------------------------------------------------------------ */
/* calc_ffn_arr + calc_ads_arr, rt_routines.f90:2186-2297. For nside <= 256 the reference reads
 * tables built by pre_calc_ads_arr (:2088-2180) from the centre of pixel `ipix` at `nside`; the
 * same numbers are recomputed here on the fly. For "reco" rays the caller passes ipix = 1
 * (main_dir_loop, rt_routines.f90:1398-1403), so the table row of pixel 1 is used: kept as is. */
static void calc_ffn_arr(const orc_ctx *c, tstate *t, double al, double dl, int ipix)
{
    double th = dl, ph = al;
    if (t->nside <= 256) orc_pix2ang_nest(t->nside, ipix, &th, &ph);
    if (c->nuniq > 0) {
        double sin_dl = sin(th), cos_dl = cos(th), sin_al = sin(ph), cos_al = cos(ph);
        for (int i = 0; i < c->nuniq; i++)
            for (int p = 0; p < c->npix_unique[i]; p++)
                t->ads[i][p] = (c->cos_theta_sca[i][p] * cos_dl + c->sin_theta_sca[i][p] * sin_dl * (c->cos_phi_sca[i][p] * cos_al + c->sin_phi_sca[i][p] * sin_al));
    }
    for (int i = 0; i < c->L; i++) {
        if (!t->iq_a[i]) continue;
        int ik = c->ik_sca_arr[i];
        if (ik != -1) {
            double g = c->gsca[i];
            double konst = (1.0 - g * g) / c->npix_hp_arr[i];
            double *f = t->ffn[ik];
            int n = c->npix_unique[ik];
            for (int p = 0; p < n; p++) f[p] = pow(1.0 + g * g - 2.0 * g * t->ads[ik][p], -1.5);
            for (int p = 0; p < n; p++) f[p] = konst * f[p];
            double tot_hg = 0;
            for (int p = 0; p < c->npix_hp_arr[i]; p++) tot_hg += f[p];
            for (int p = 0; p < n; p++) f[p] = f[p] / tot_hg;
            memcpy(t->ffn_mw[i], f, n * sizeof(double));
        } else {
            for (int p = 0; p < c->npix_arr[i]; p++) t->ffn_mw[i][p] = (double)(1 / c->npix_hp_arr[i]);
        }
    }
}

/* check_wall_hit, rt_routines.f90:2572-2596 */
static int check_wall_hit(const orc_ctx *c, int cc)
{
    if (c->rt_type == ORC_RTT_DIR_SRC) return 0;
    int hit = 0;
    for (int k = 0; k < 3; k++)
        if (c->wall_on[k] && (CC(c, cc, k) < c->wall_coord[k][0] || CC(c, cc, k) > c->wall_coord[k][1])) hit = 1;
    return hit;
}

/* process_scatt_rad, rt_routines.f90:2638-2726 (no-communication, sequential-scattering branch) */
static void process_scatt_rad(orc_ctx *c, tstate *t, int cc)
{
    double s = 0;
    for (int il = 0; il < c->L; il++) if (t->iq_a[il]) s += t->en_sca[il];
    if (s == 0.) return;
    for (int il = 0; il < c->L; il++) {
        if (!t->iq_a[il]) continue;
        double e = t->en_sca[il];
        int n = c->npix_arr[il];
        double *dst = c->scaspe[il] + (size_t)cc * n;
        const double *f = t->ffn_mw[il];
        if (c->nthreads > 1) for (int p = 0; p < n; p++) atomic_add(&dst[p], e * f[p]);
        else for (int p = 0; p < n; p++) dst[p] = dst[p] + e * f[p];
    }
}

/* deposit, rt_routines.f90:2362-2557 */
static void deposit(orc_ctx *c, tstate *t, int cc, double length, int nc, double psel, int *ray_status, int flag_beam)
{
    const int L = c->L, rt = c->rt_type;
    double *I = t->ray_intensity, *tau_c = t->tau_c, *ex = t->exp_tau_c, *i_av = t->i_av;
    const double *dens = c->dens + (size_t)cc * L;
    int npix_beam = 12 * (t->nside * t->nside);
    for (int il = 0; il < L; il++) if (t->iq_a[il]) { tau_c[il] = dens[il] * length; ex[il] = exp(-tau_c[il]); }
    int clvl = c->lvl[nc];
    double csize_i = c->csize_arr[clvl], area_i = c->carea_arr[clvl];
    double beam_i = 4 * PI_ / npix_beam;
    clvl = c->lvl[cc];
    double csize_f = c->csize_arr[clvl], vol_f = c->cvol_arr[clvl];
    double beam_f = c->carea_arr[clvl] / (psel * psel);

    if (cc != nc) {
        for (int il = 0; il < L; il++) if (t->iq_a[il]) {
            if (tau_c[il] < TAU_THIN || I[il] == 0) i_av[il] = I[il];
            else i_av[il] = I[il] / tau_c[il] * (1 - ex[il]);
        }
    } else {
        for (int il = 0; il < L; il++) if (t->iq_a[il]) {
            if (tau_c[il] < TAU_THIN || I[il] == 0) i_av[il] = I[il];
            else if (rt != ORC_RTT_PRECALC_SRC && rt != ORC_RTT_DIR_SRC) {
                tau_c[il] = dens[il] * csize_i;
                ex[il] = exp(-tau_c[il]);
                i_av[il] = I[il] / (tau_c[il] * tau_c[il]) * (ex[il] + tau_c[il] - 1);
            } else {
                i_av[il] = I[il] / (dens[il] * length) * (1 - ex[il]);
            }
        }
    }

    if (rt == ORC_RTT_PRECALC_SRC || rt == ORC_RTT_PRECALC_CELL) {
        if (psel > c->rad_lim * c->modelsize) { *ray_status = RAS_GONE; return; }
        if ((beam_f / c->bm_par <= beam_i) && !flag_beam) { *ray_status = RAS_GO_HIGH; return; }
        if ((beam_f / c->bm_par_max < beam_i) || (t->nside <= 2 * NSIDE_MIN) || flag_beam) {
            double *u = c->u_fest + (size_t)cc * L;
            for (int il = 0; il < L; il++) if (t->iq_a[il]) {
                double v = i_av[il] * length / c->cs / vol_f * area_i * beam_i;
                if (c->nthreads > 1) atomic_add(&u[il], v); else u[il] = u[il] + v;
            }
        } else { *ray_status = RAS_GO_LOW; return; }
    } else if (rt == ORC_RTT_DIR_SRC || rt == ORC_RTT_DIR_CELL || rt == ORC_RTT_SCATT) {
        if ((beam_f / c->bm_par <= beam_i) && !flag_beam) {
            const double *uf = c->u_fest + (size_t)cc * L;
            int any = 0;
            for (int il = 0; il < L; il++) if (t->iq_a[il]) {
                t->u_beam[il] = i_av[il] * csize_f / c->cs / vol_f * area_i * beam_f;
                if (t->u_beam[il] < c->en_lim * uf[il]) {
                    I[il] = 0;
                    t->lum_lost_temp[il] = t->lum_lost_temp[il] + i_av[il] * beam_i * csize_i * csize_i;
                }
                if (I[il] > 0) any = 1;
            }
            int wall_hit = check_wall_hit(c, cc);
            if (any && !wall_hit) { *ray_status = RAS_GO_HIGH; return; }
            *ray_status = RAS_GONE;
            return;
        } else if ((beam_f / c->bm_par_max < beam_i) || (t->nside <= 2 * NSIDE_MIN) || flag_beam) {
            double *u = c->u_final + (size_t)cc * L;
            for (int il = 0; il < L; il++) if (t->iq_a[il]) {
                double v = i_av[il] * length / c->cs / vol_f * area_i * beam_i;
                if (c->nthreads > 1) atomic_add(&u[il], v); else u[il] = u[il] + v;
            }
            for (int il = 0; il < L; il++) if (t->iq_a[il]) {
                double k = c->ksca_norm[il];
                if (cc != nc) {
                    if (tau_c[il] > TAU_THIN) t->en_sca[il] = I[il] * (1.0 - ex[il]) * k * area_i * beam_i;
                    else t->en_sca[il] = I[il] * tau_c[il] * k * area_i * beam_i;
                } else {
                    if (tau_c[il] > TAU_THIN) t->en_sca[il] = I[il] / (tau_c[il]) * (ex[il] + tau_c[il] - 1) * k * area_i * beam_i;
                    else t->en_sca[il] = I[il] * tau_c[il] / 2 * k * area_i * beam_i;
                }
            }
            process_scatt_rad(c, t, cc);
        } else { *ray_status = RAS_GO_LOW; return; }
    }
    /* update ray_intensity, rt_routines.f90:2543-2555 */
    if (cc != nc) {
        for (int il = 0; il < L; il++) if (t->iq_a[il]) I[il] = I[il] * ex[il];
    } else {
        for (int il = 0; il < L; il++) if (t->iq_a[il]) {
            if (!(tau_c[il] < TAU_THIN)) I[il] = I[il] / (tau_c[il]) * (1 - ex[il]);
        }
    }
}

/* ray_tracing, rt_routines.f90:1618-1857 */
static void ray_tracing(orc_ctx *c, tstate *t, int nc, double al, double dl, int clvl, const double *src_lum,
                        double prev_ray, int isel_ray, int ray_type, int ipix, int cc_old)
{
    const int L = c->L, rt = c->rt_type;
    double cproj[3];
    int incvec[3], cc, isel = 0, isel_old, ray_status, flag_beam, cc_new;
    double prev, psel = 0;
    t->cnt.rays++;
    if (c->trace) fprintf(c->trace, "%d %d %d %d %d\n", nc, t->nside, (ray_type == RAY_RECO) ? orc_ang2pix_nest(t->nside, dl, al) : ipix, cc_old, ray_type);
    /* calc_cproj, rt_routines.f90:1945-1953 */
    cproj[0] = sin(dl) * cos(al);
    cproj[1] = sin(dl) * sin(al);
    cproj[2] = cos(dl);
    if (cc_old == nc) {
        cc = nc; prev = 0.0; ray_status = RAS_FIRST; isel_old = 0; flag_beam = 1; t->clvl_old = -1;
        memcpy(t->ccindd, t->ccindd_nc, sizeof(int) * 3 * c->max_lvl);
        cc_new = -1;
    } else {
        int out;
        if (!find_cc_new2(c, t, cc_old, &cc_new, cproj, prev_ray, &out, &clvl)) return;
        prev = prev_ray;
        if (out) return;
        cc = cc_new; ray_status = RAS_RELAUNCH; isel_old = isel_ray; flag_beam = 1; t->clvl_old = -1;
    }
    for (int k = 0; k < 3; k++) {
        int v = (int)(cproj[k] / GLEPSILON);
        if (v > 1) v = 1;
        if (v < -1) v = -1;
        incvec[k] = v;
    }
    for (int k = 0; k < 3; k++) cproj[k] = 1 / cproj[k];
    double *I = t->ray_intensity;
    double norm = 4 * PI_ * c->csize[nc] * c->csize[nc];
    for (int il = 0; il < L; il++) I[il] = src_lum[il] / norm;
    /* set_iq_list, rt_routines.f90:1860-1893: the active set is fixed at launch (sticky) */
    for (int il = 0; il < L; il++) t->iq_a[il] = (I[il] > 0);
    if (rt == ORC_RTT_DIR_SRC || rt == ORC_RTT_DIR_CELL || rt == ORC_RTT_SCATT) calc_ffn_arr(c, t, al, dl, ipix);

    int dbg = 0;
    if (getenv("ORC_DEBUG_RAY")) {
        int dn = 0, dp = 0;
        sscanf(getenv("ORC_DEBUG_RAY"), "%d:%d", &dn, &dp);
        dbg = (t->nside == dn && orc_ang2pix_nest(t->nside, dl, al) == dp);
        if (dbg) fprintf(stderr, "RAY nside %d pix %d type %d nc %d cc_old %d start cc %d clvl %d prev %.17g dir %.17g %.17g %.17g\n", t->nside, dp, ray_type, nc, cc_old, cc, clvl, prev, 1 / cproj[0], 1 / cproj[1], 1 / cproj[2]);
    }
    for (;;) {
        t->cnt.intersections++;
        calc_psel(c, t, incvec, &psel, &isel, clvl, cc, cproj);
        if (dbg) {
            double beam_f = c->carea_arr[c->lvl[cc]] / (psel * psel), beam_i = 4 * PI_ / (12.0 * t->nside * t->nside);
            fprintf(stderr, "   cc %d lvl %d ccoord %.6f %.6f %.6f psel %.17g isel %d len %.3e flag_beam %d beam_f/bm %.6g beam_f/bmmax %.6g beam_i %.6g I0 %.6g\n", cc, clvl, CC(c, cc, 0), CC(c, cc, 1), CC(c, cc, 2), psel, isel, psel - prev,
                    flag_beam, beam_f / c->bm_par, beam_f / c->bm_par_max, beam_i, t->ray_intensity[0]);
        }
        double length = psel - prev;
        if (fabs(length) < LEN_EPS * c->csize_arr[clvl]) length = 0;
        if (length < 0) { fprintf(stderr, "oracle: negative length %g in ray_tracing (cc=%d nc=%d)\n", length, cc, nc); t->error = 1; return; }
        if (cc != nc && cc != cc_new) flag_beam = 0;
        if (length != 0.) {
            t->cnt.deposits++;
            deposit(c, t, cc, length, nc, psel, &ray_status, flag_beam);
        }
        if (ray_status == RAS_GO_HIGH) {
            for (int il = 0; il < L; il++) t->lum_tmp[il] = I[il] * norm;
            double dplane_ray = prev / cproj[isel_old - 1];
            add_to_high(t, dl, al, t->nside, prev, isel_old, dplane_ray, t->lum_tmp, cc, RAY_HIGH);
            return;
        } else if (ray_status == RAS_GONE) {
            t->ipsel_av++; t->psel_av += psel;   /* :1746-1748 */
            return;
        } else if (ray_status == RAS_GO_LOW) {
            for (int il = 0; il < L; il++) t->lum_tmp[il] = I[il] * norm;
            add_to_low(t, dl, al, t->nside, prev, isel_old, cproj, t->lum_tmp, cc);
            return;
        }
        prev = psel;
        isel_old = isel;
        t->clvl_old = clvl;
        /* next cell, rt_routines.f90:1778-1841 */
        for (int i = 0; i < 3; i++) {
            if (i + 1 == isel) {
                int k = incvec[i];
                for (int j = clvl; j >= 1; j--) {
                    increment(c, j, &t->ccindd[(j - 1) * 3 + i], k, &k);
                    if (k == 0) break;
                }
                if (k != 0) { /* leaving the model */
                    t->ipsel_av++; t->psel_av += psel;   /* :1792-1793 */
                    if (rt == ORC_RTT_I_OBS_DIR_CELL || rt == ORC_RTT_I_OBS_DIR_SRC || rt == ORC_RTT_I_OBS)
                        for (int il = 0; il < L; il++) t->i_obs_temp[il] = I[il];
                    return;
                }
                for (int j = clvl + 1; j <= c->max_lvl; j++) {
                    int b = c->base[(j > 1) ? 1 : 0];
                    if (incvec[i] == 1) t->ccindd[(j - 1) * 3 + i] = 0;
                    else if (incvec[i] == -1) t->ccindd[(j - 1) * 3 + i] = b - 1;
                }
            } else {
                double cellsize = c->csize_arr[clvl];
                double b = psel / cproj[i] + (t->src_ccoord[i] - CC(c, cc, i) + 0.5 * cellsize);
                for (int j = clvl + 1; j <= c->max_lvl; j++) {
                    int bb = c->base[(j > 1) ? 1 : 0];
                    cellsize = c->csize_arr[j];
                    int v = (int)(b / cellsize);
                    if (v == bb) v = bb - 1;
                    t->ccindd[(j - 1) * 3 + i] = v;
                    b = fmod(b, cellsize);
                }
            }
        }
        if (!ccindd_to_cc(c, t->ccindd, &cc, &clvl)) { fprintf(stderr, "oracle: next cell not found (ray_tracing)\n"); t->error = 1; return; }
    }
}

/* assign_src_lum, rt_routines.f90:3043-3071 */
static void assign_src_lum(const orc_ctx *c, tstate *t, double theta, double phi, double *src_lum)
{
    int ip_arr[8];
    for (int i = 0; i < c->nuniq; i++) ip_arr[i] = orc_ang2pix_nest(1 << c->kp_unique[i], theta, phi);
    for (int il = 0; il < c->L; il++) {
        int ik = c->ik_sca_arr[il];
        int ip = (ik != -1) ? ip_arr[ik] : 0;
        src_lum[il] = t->scaspe_temp[il][ip] * c->npix_hp_arr[il];
    }
}

/* src_lum_value, rt_routines.f90:1897-1919 */
static void src_lum_value(const orc_ctx *c, int id, double *out)
{
    int rt = c->rt_type, L = c->L;
    if (rt == ORC_RTT_PRECALC_SRC || rt == ORC_RTT_DIR_SRC || rt == ORC_RTT_I_OBS_DIR_SRC) {
        if (c->src_cell[id] != -1) memcpy(out, c->p_src_lum + (size_t)c->src_cell[id] * L, L * sizeof(double));
        else memset(out, 0, L * sizeof(double));
    } else {
        memcpy(out, c->lumcell + (size_t)id * L, L * sizeof(double));
    }
}

/* main_dir_loop, rt_routines.f90:1343-1467 */
static void main_dir_loop(orc_ctx *c, tstate *t, int src_id, double *src_lum)
{
    const int L = c->L, rt = c->rt_type;
    for (int idir = 0; idir < NPIX_MAIN; idir++) {
        for (int idir2 = 0; idir2 < 4; idir2++) {
            t->nside = NSIDE_MIN;
            memset(t->lum_lost_temp, 0, L * sizeof(double));
            t->low.n = 0;
            create_high_ray_list(t, idir, idir2, src_lum, src_id, t->nside / 2);
            for (;;) {
                extract_ray_list(t, t->nside);
                if (t->nside > t->cnt.max_nside) t->cnt.max_nside = t->nside;
                int next = t->no_ray ? 0 : t->ext.n;
                for (int ie = 0; ie < next; ie++) {
                    rayrec e = t->ext.r[ie]; /* copy: the queues may grow while tracing */
                    memcpy(t->src_lum_ray, t->ext.lum + (size_t)ie * L, L * sizeof(double));
                    double theta = e.theta, phi = e.phi;
                    int i0 = 0, i1 = -1;
                    if (e.ray_type == RAY_HIGH) { i0 = orc_ang2pix_nest(e.nside, theta, phi) * 4; i1 = i0 + 3; }
                    if (e.ray_type == RAY_RECO) { i0 = 1; i1 = 1; }
                    if (e.ray_type == RAY_LOW) { i0 = orc_ang2pix_nest(t->nside, theta, phi); i1 = i0; }
                    if (e.ray_type == RAY_GONE) continue;
                    for (int ipix = i0; ipix <= i1; ipix++) {
                        if (e.ray_type == RAY_HIGH) orc_pix2ang_nest(t->nside, ipix, &theta, &phi);
                        if (rt == ORC_RTT_SCATT && e.cc_old == src_id) {
                            assign_src_lum(c, t, theta, phi, src_lum);
                            memcpy(t->src_lum_ray, src_lum, L * sizeof(double));
                        }
                        int clvl = c->lvl[src_id];
                        ray_tracing(c, t, src_id, phi, theta, clvl, t->src_lum_ray, e.prev, e.isel, e.ray_type, ipix, e.cc_old);
                        if (t->error) return;
                    }
                }
                if (define_next_level(t)) break;
                if (t->nside > 8192 || t->nside < 1) { fprintf(stderr, "oracle: nside out of range\n"); t->error = 1; return; }
            }
            if (rt == ORC_RTT_DIR_SRC || rt == ORC_RTT_DIR_CELL || rt == ORC_RTT_SCATT)
                for (int il = 0; il < L; il++) atomic_add(&c->lum_lost[il], t->lum_lost_temp[il]);
        }
    }
}

static void merge_counters(orc_ctx *c, const tstate *t)
{
#ifdef _OPENMP
#pragma omp critical
#endif
    {
        c->cnt.intersections += t->cnt.intersections; c->cnt.rays += t->cnt.rays; c->cnt.deposits += t->cnt.deposits;
        if (t->cnt.max_high > c->cnt.max_high) c->cnt.max_high = t->cnt.max_high;
        if (t->cnt.max_low > c->cnt.max_low) c->cnt.max_low = t->cnt.max_low;
        if (t->cnt.max_ext > c->cnt.max_ext) c->cnt.max_ext = t->cnt.max_ext;
        if (t->cnt.max_nside > c->cnt.max_nside) c->cnt.max_nside = t->cnt.max_nside;
        if (t->error) c->error = 1;
    }
}

/* rt_loop, rt_routines.f90:985-1058 */
void orc_rt_loop(orc_ctx *c, int rt_type)
{
    c->rt_type = rt_type;
    const int L = c->L;
    int nim = c->N / c->np + 1;
    memset(c->psel_av_last, 0, (size_t)(c->N + c->nsrc) * sizeof(double));
#ifdef _OPENMP
#pragma omp parallel num_threads(c->nthreads)
#endif
    {
        tstate t;
        ts_init(c, &t);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int im = 0; im < nim; im++) {
            int i = im * c->np + c->id;
            if (i > c->N - 1 || t.error) continue;
            src_lum_value(c, i, t.src_lum);
            double s = 0;
            for (int il = 0; il < L; il++) s += t.src_lum[il];
            if (c->cchild[i] != -1 || s == 0) continue;
            /* calc_src_ccoord, rt_routines.f90:1922-1939 */
            if (rt_type == ORC_RTT_PRECALC_SRC || rt_type == ORC_RTT_DIR_SRC)
                for (int k = 0; k < 3; k++) t.src_ccoord[k] = c->p_src_pos[(size_t)c->src_cell[i] * 3 + k];
            else
                for (int k = 0; k < 3; k++) t.src_ccoord[k] = CC(c, i, k);
            cindex_to_ccindd(c, i, c->lvl[i], t.ccindd_nc);
            /* assign_scaspe_temp_arr, rt_routines.f90:1143-1170 */
            if (rt_type == ORC_RTT_SCATT)
                for (int il = 0; il < L; il++)
                    memcpy(t.scaspe_temp[il], c->scaspe_prev[il] + (size_t)i * c->npix_arr[il], c->npix_arr[il] * sizeof(double));
            t.ipsel_av = 0; t.psel_av = 0;   /* :1020-1021 */
            main_dir_loop(c, &t, i, t.src_lum);
            {   /* :1456-1465 (the reference stores it for the dir and scattering stages when print_psel_av) */
                const int sid = (rt_type == ORC_RTT_PRECALC_SRC || rt_type == ORC_RTT_DIR_SRC) ? c->N + c->src_cell[i] : i;
                c->psel_av_last[sid] = t.ipsel_av ? t.psel_av / t.ipsel_av : 0.0;
            }
        }
        merge_counters(c, &t);
        ts_free(c, &t);
    }
}

/* rt_loop_iobs, rt_routines.f90:1470-1603 (external observers only) */
void orc_rt_loop_iobs(orc_ctx *c, int rt_type)
{
    c->rt_type = rt_type;
    const int L = c->L;
    if (c->ndir == 0) return;
#ifdef _OPENMP
#pragma omp parallel num_threads(c->nthreads)
#endif
    {
        tstate t;
        ts_init(c, &t);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
        for (int i = 0; i < c->N; i++) {
            if (t.error) continue;
            src_lum_value(c, i, t.src_lum_ray);
            double s = 0;
            for (int il = 0; il < L; il++) s += t.src_lum_ray[il];
            if (c->cchild[i] != -1 || s == 0) continue;
            if (rt_type == ORC_RTT_I_OBS_DIR_SRC)
                for (int k = 0; k < 3; k++) t.src_ccoord[k] = c->p_src_pos[(size_t)c->src_cell[i] * 3 + k];
            else
                for (int k = 0; k < 3; k++) t.src_ccoord[k] = CC(c, i, k);
            int sid = (rt_type == ORC_RTT_I_OBS_DIR_SRC) ? c->N + c->src_cell[i] : i;
            cindex_to_ccindd(c, i, c->lvl[i], t.ccindd_nc);
            if (rt_type == ORC_RTT_I_OBS)
                for (int il = 0; il < L; il++)
                    memcpy(t.scaspe_temp[il], c->scaspe_tot[il] + (size_t)i * c->npix_arr[il], c->npix_arr[il] * sizeof(double));
            for (int j = 0; j < c->ndir; j++) {
                memset(t.i_obs_temp, 0, L * sizeof(double));
                double theta = c->dir_out[j], phi = c->dir_out[c->ndir + j];
                if (rt_type == ORC_RTT_I_OBS) {
                    if (!c->rta_i_obs) {
                        for (int il = 0; il < L; il++)
                            t.src_lum_ray[il] = t.scaspe_temp[il][c->npix_hp_arr[il] + j] * c->npix_hp_arr[il];
                    } else {  /* :1544-1546: the HEALPix pixel that holds the direction */
                        assign_src_lum(c, &t, theta, phi, t.src_lum_ray);
                    }
                }
                t.nside = NSIDE_MIN; /* `nside` keeps its last value in the reference; unused by the i_obs branch of deposit */
                ray_tracing(c, &t, i, phi, theta, c->lvl[i], t.src_lum_ray, 0.0, 0, RAY_I_OBS, 0, i);
                double *dst = c->i_obs + ((size_t)j * (c->N + c->nsrc) + sid) * L;
                for (int il = 0; il < L; il++) dst[il] = dst[il] + t.i_obs_temp[il];
            }
        }
        merge_counters(c, &t);
        ts_free(c, &t);
    }
}

/* reduce_u_final_arr for one rank, io_routines.f90:3732-3761 */
void orc_rt_mpi(orc_ctx *c, int rt_type)
{
    if (rt_type != ORC_RTT_SCATT) return;
    size_t NL = (size_t)c->N * c->L;
    for (size_t k = 0; k < NL; k++) {
        double d = c->u_final[k] - c->u_fest[k];
        if (d < 0) d = 0;
        c->u_final[k] = d + c->u_fest[k];
    }
}

/* rt_prepare case rtt_scatt (dartray_hub.f90:432-457) + calc_total_luminosity_sca (rt_routines.f90:663-758) */
int orc_scatt_prepare(orc_ctx *c)
{
    const int L = c->L;
    size_t NL = (size_t)c->N * L;
    c->rt_type = ORC_RTT_SCATT;
    memcpy(c->u_fest, c->u_final, NL * sizeof(double));
    memcpy(c->lum_lost_prev, c->lum_lost, L * sizeof(double));
    c->iterations++;
    if (c->iterations == 1) {
        for (int il = 0; il < L; il++) {
            size_t n = (size_t)c->N * c->npix_arr[il];
            c->scaspe_tot[il] = (double *)malloc(n * sizeof(double));
            memcpy(c->scaspe_tot[il], c->scaspe[il], n * sizeof(double)); /* create_scaspe_tot, :3012-3040 */
            c->scaspe_prev[il] = (double *)malloc(n * sizeof(double));    /* create_scaspe_prev, :4049-4080 */
        }
        c->bm_par = c->bm_par_sca;
    }
    for (int il = 0; il < L; il++) memcpy(c->scaspe_prev[il], c->scaspe[il], (size_t)c->N * c->npix_arr[il] * sizeof(double));
    /* calc_total_luminosity_sca */
    if (c->iterations == 1) {
        for (int il = 0; il < L; il++) {
            double s = 0;
            int n = c->npix_arr[il], nhp = c->npix_hp_arr[il];
            for (int i = 0; i < c->N; i++) for (int p = 0; p < nhp; p++) s += c->scaspe[il][(size_t)i * n + p];
            c->tot_rad_en[il] = s;
        }
    }
    memset(c->lumcell, 0, NL * sizeof(double));
    for (int i = 0; i < c->N; i++) {
        if (c->cchild[i] != -1) continue;
        for (int il = 0; il < L; il++) {
            int n = c->npix_arr[il], nhp = c->npix_hp_arr[il];
            double s = 0;
            for (int p = 0; p < nhp; p++) s += c->scaspe[il][(size_t)i * n + p];
            c->lumcell[(size_t)i * L + il] = s;
        }
    }
    double sum_tot = 0;
    for (int il = 0; il < L; il++) sum_tot += c->tot_rad_en[il];
    if (sum_tot > 0) {
        int all = 1;
        for (int il = 0; il < L; il++) {
            double tl = 0;
            for (int i = 0; i < c->N; i++) tl += c->lumcell[(size_t)i * L + il];
            if (!(tl / c->tot_rad_en[il] < c->conv_en_lim)) all = 0;
        }
        c->cnflag = all;
    } else c->cnflag = 1;
    if (c->max_sca_iterations >= 0 && c->iterations > c->max_sca_iterations) c->cnflag = 1;
    /* sequential scattering bookkeeping, rt_routines.f90:747-756 */
    for (int il = 0; il < L; il++) {
        size_t n = (size_t)c->N * c->npix_arr[il];
        if (c->iterations > 1) for (size_t k = 0; k < n; k++) c->scaspe_tot[il][k] = c->scaspe_tot[il][k] + c->scaspe[il][k];
        memset(c->scaspe[il], 0, n * sizeof(double));
    }
    return c->cnflag;
}

/* calc_total_luminosity_sca, case rtt_i_obs, rt_routines.f90:760-771 */
void orc_iobs_prepare(orc_ctx *c)
{
    const int L = c->L;
    memset(c->lumcell, 0, (size_t)c->N * L * sizeof(double));
    for (int i = 0; i < c->N; i++) {
        if (c->cchild[i] != -1) continue;
        for (int il = 0; il < L; il++) {
            int n = c->npix_arr[il];
            double s = 0;
            for (int p = 0; p < n; p++) s += c->scaspe_tot[il][(size_t)i * n + p];
            c->lumcell[(size_t)i * L + il] = s;
        }
    }
}

/* dartray_main, dartray_hub.f90:12-67 */
void orc_run_all(orc_ctx *c, int stages)
{
    if (stages & 1) { orc_rt_loop(c, ORC_RTT_PRECALC_CELL); orc_rt_loop(c, ORC_RTT_PRECALC_SRC); }
    if (stages & 2) { orc_rt_loop(c, ORC_RTT_DIR_CELL); orc_rt_loop(c, ORC_RTT_DIR_SRC); orc_rt_mpi(c, ORC_RTT_DIR_SRC); }
    if (stages & 4) { orc_rt_loop_iobs(c, ORC_RTT_I_OBS_DIR_CELL); orc_rt_loop_iobs(c, ORC_RTT_I_OBS_DIR_SRC); }
    if (stages & 8) {
        for (;;) {
            if (orc_scatt_prepare(c)) break;
            orc_rt_loop(c, ORC_RTT_SCATT);
            orc_rt_mpi(c, ORC_RTT_SCATT);
            if (c->error) break;
        }
        if (stages & 16) { orc_iobs_prepare(c); orc_rt_loop_iobs(c, ORC_RTT_I_OBS); }
    }
}

/* sum_i_obs for 'erg/s/Hz/pc^2/sr', sed_routines.f90:211-246; sed[ndir][L] */
void orc_sed(orc_ctx *c, double *sed)
{
    const double parsec = (double)3.08567758E16f;
    const int L = c->L, M = c->N + c->nsrc;
    for (int il = 0; il < L; il++) for (int j = 0; j < c->ndir; j++) {
        double s = 0;
        for (int i = 0; i < c->N; i++) {
            double v = c->i_obs[((size_t)j * M + i) * L + il];
            v = v * pow(10.0, -7); v = v * pow(10.0, 26) / pow(parsec, 2.);
            double r = c->csize[i] / c->dist_obs;
            s += v * (r * r);
        }
        double s2 = 0;
        for (int k = 0; k < c->nsrc; k++) {
            double v = c->i_obs[((size_t)j * M + c->N + k) * L + il];
            v = v * pow(10.0, -7); v = v * pow(10.0, 26) / pow(parsec, 2.);
            double r = c->csize[c->cell_src[k]] / c->dist_obs;
            s2 += v * (r * r);
        }
        sed[(size_t)j * L + il] = s + s2;
    }
}

/* developer aid: (re)direct the ray-launch trace; NULL closes it */
void orc_trace_open(orc_ctx *c, const char *path)
{
    if (c->trace) fclose(c->trace);
    c->trace = path ? fopen(path, "w") : NULL;
}

int orc_ncell(const orc_ctx *c) { return c->N; }
int orc_lnum(const orc_ctx *c) { return c->L; }
int orc_npix(const orc_ctx *c, int il) { return c->npix_arr[il]; }
int orc_npix_hp(const orc_ctx *c, int il) { return c->npix_hp_arr[il]; }
int orc_kp(const orc_ctx *c, int il) { return c->kp_sca_arr[il]; }
int orc_iterations(const orc_ctx *c) { return c->iterations; }
double orc_en_lim(const orc_ctx *c) { return c->en_lim; }
double *orc_u_fest(orc_ctx *c) { return c->u_fest; }
double *orc_u_final(orc_ctx *c) { return c->u_final; }
double *orc_scaspe(orc_ctx *c, int il) { return c->scaspe[il]; }
double *orc_scaspe_tot(orc_ctx *c, int il) { return c->scaspe_tot[il]; }
double *orc_scaspe_prev(orc_ctx *c, int il) { return c->scaspe_prev[il]; }
double *orc_lumcell(orc_ctx *c) { return c->lumcell; }
double *orc_lum_lost(orc_ctx *c) { return c->lum_lost; }
double *orc_lum_lost_prev(orc_ctx *c) { return c->lum_lost_prev; }
double *orc_i_obs(orc_ctx *c) { return c->i_obs; }
int32_t *orc_src_cell(orc_ctx *c) { return c->src_cell; }
double *orc_p_src_pos(orc_ctx *c) { return c->p_src_pos; }
void orc_get_counters(const orc_ctx *c, orc_counters *out) { *out = c->cnt; }
void orc_reset_counters(orc_ctx *c) { memset(&c->cnt, 0, sizeof(c->cnt)); }
