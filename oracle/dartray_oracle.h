/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99, double precision) of DART-Ray's ray-cell radiative-transfer hot
 * path: rt_loop -> main_dir_loop -> ray_tracing -> deposit/process_scatt_rad of
 * /root/reference/rt_routines.f90 with the ray queues of ray_list.f90, the HEALPix NESTED helpers
 * of healpix_routines.f90 and the tree helpers of smooth_grid_routines.f90. Every function in
 * dartray_oracle.c cites the reference file:line it follows.
 *
 * Parity status: PINNED. The restatement is checked (tests/test_oracle_golden.py) against golden
 * vectors produced by the unmodified reference modules compiled with amdflang and driven by
 * oracle/ref_driver.f90 (tests/golden/, generator oracle/make_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (dart-ray_amd/) never links, imports or falls back to it.
 */
#ifndef DARTRAY_ORACLE_H
#define DARTRAY_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* rt_type ids, rt_routines.f90:195-217 */
enum {
    ORC_RTT_PRECALC_CELL = 1, ORC_RTT_PRECALC_SRC = 2, ORC_RTT_DIR_CELL = 5,
    ORC_RTT_I_OBS_DIR_CELL = 6, ORC_RTT_DIR_SRC = 7, ORC_RTT_I_OBS_DIR_SRC = 8,
    ORC_RTT_SCATT = 10, ORC_RTT_I_OBS = 11
};

typedef struct orc_ctx orc_ctx;

typedef struct {
    int64_t intersections; /* passes through the main: loop body, rt_routines.f90:1689 */
    int64_t rays;          /* calls of ray_tracing, rt_routines.f90:1618 */
    int64_t deposits;      /* intersections with length /= 0, rt_routines.f90:1731 */
    int64_t max_high, max_low, max_ext; /* largest queue sizes seen (sizing of GPU arenas) */
    int max_nside;
} orc_counters;

/* Create a context from a case file written by dart-ray_amd/grids.py (Case.write). */
orc_ctx *orc_create_from_file(const char *path);
/* Create from arrays (all copied). ccoord [N][3], dens/dens_stars [N][L], p_src_pos [nsrc][3],
 * p_src_lum [nsrc][L], dir_out [2][ndir] (thetas then phis), wall_coord [3][2] in model units. */
orc_ctx *orc_create(int ncell, int max_lvl, const int *base, int lnum, int nsrc, int ndir,
                    int kp_sca_max, int bm_par, int bm_par_sca, int bm_par_max,
                    const int *wall_on, const double *wall_coord, int max_sca_iterations,
                    double modelsize, double rad_lim, double accuracy, double conv_en_lim,
                    double dist_obs, double cs,
                    const int32_t *cchild, const int32_t *lvl, const int64_t *cindex,
                    const double *ccoord, const double *csize, const double *dens,
                    const double *dens_stars, const double *ksca_norm, const double *gsca,
                    const double *p_src_pos, const double *p_src_lum, const double *dir_out);
void orc_destroy(orc_ctx *c);

/* MPI-like source decomposition of rt_loop (rt_routines.f90:1003-1005) and thread count. */
void orc_set_parallel(orc_ctx *c, int np, int id, int nthreads);

/* Stage entry points (same names as the reference's). */
void orc_rt_loop(orc_ctx *c, int rt_type);
void orc_rt_loop_iobs(orc_ctx *c, int rt_type);
/* rt_mpi for one rank: the u_final - u_fest clamp of reduce_u_final_arr (io_routines.f90:3743-3757) */
void orc_rt_mpi(orc_ctx *c, int rt_type);
/* rt_prepare, case rtt_scatt (dartray_hub.f90:432-457) + calc_total_luminosity_sca; returns cnflag */
int orc_scatt_prepare(orc_ctx *c);
/* rt_prepare, case rtt_i_obs: calc_total_luminosity_sca for rtt_i_obs (rt_routines.f90:760-771) */
void orc_iobs_prepare(orc_ctx *c);
/* The whole dartray_main sequence (dartray_hub.f90:12-67), stages bitmask as in ref_driver.f90 */
void orc_run_all(orc_ctx *c, int stages);

/* State access. Arrays are owned by the context. */
int orc_ncell(const orc_ctx *c);
int orc_lnum(const orc_ctx *c);
int orc_npix(const orc_ctx *c, int il);    /* npix_arr  */
int orc_npix_hp(const orc_ctx *c, int il); /* npix_hp_arr */
int orc_kp(const orc_ctx *c, int il);      /* kp_sca_arr */
int orc_iterations(const orc_ctx *c);
double orc_en_lim(const orc_ctx *c);
double *orc_u_fest(orc_ctx *c);            /* [N][L] */
double *orc_u_final(orc_ctx *c);           /* [N][L] */
double *orc_scaspe(orc_ctx *c, int il);    /* [N][npix] */
double *orc_scaspe_tot(orc_ctx *c, int il);
double *orc_scaspe_prev(orc_ctx *c, int il);
double *orc_lumcell(orc_ctx *c);           /* [N][L] */
double *orc_lum_lost(orc_ctx *c);          /* [L] */
double *orc_lum_lost_prev(orc_ctx *c);     /* [L] */
double *orc_i_obs(orc_ctx *c);             /* [ndir][N+nsrc][L] */
int32_t *orc_src_cell(orc_ctx *c);         /* [N] */
double *orc_p_src_pos(orc_ctx *c);         /* [nsrc][3] after the tol_p nudge */
void orc_sed(orc_ctx *c, double *sed);     /* [ndir][L], sum_i_obs (sed_routines.f90:211-246) */
void orc_get_counters(const orc_ctx *c, orc_counters *out);
void orc_reset_counters(orc_ctx *c);

void orc_trace_open(orc_ctx *c, const char *path); /* developer aid: log every ray launch */

/* Unit-level entry points for tests. */
void orc_pix2ang_nest(int nside, int ipix, double *theta, double *phi);
int orc_ang2pix_nest(int nside, double theta, double phi);
int orc_locate(const orc_ctx *c, const int *ccindd, int *clvl); /* ccindd_to_cc; -1 if not found */
void orc_cindex_to_ccindd(const orc_ctx *c, int cell, int *ccindd /* [max_lvl][3] */);

#ifdef __cplusplus
}
#endif
#endif
