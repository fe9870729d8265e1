/*
 * dartray_rt.h — C ABI of the MI355X-native DART-Ray ray-cell hot path (libdartray_rt.so).
 *
 * The reference has no FFI for this path: it sits behind three argument-less Fortran module
 * procedures called from dartray_hub.f90 — rt_loop (dartray_hub.f90:22,24,31,33,49),
 * rt_loop_iobs (:37,39,55) and rt_mpi (:25,34,50) — plus the set-up helpers of rt_prepare
 * (:390-457), all communicating through module variables of smooth_grid_routines.f90 and
 * rt_routines.f90. This header is the boundary a Fortran facade binds with ISO_C_BINDING
 * (dart-ray_amd/fortran/dartray_gpu.f90): every entry point names the reference
 * procedure or module state it stands for. Plain pointers and sizes only; all arrays are HOST
 * arrays in the reference's own (Fortran, column-major) layout and are copied in/out — the library
 * never frees or reallocates caller memory (SURVEY.md section 8b "Ownership").
 *
 * Error convention: every function returns DRT_OK (0) or a negative DRT_E_* code; the facade turns
 * a non-zero status into `call stop_prc` (smooth_grid_routines.f90:869-892). drt_last_error()
 * gives the message. There is NO CPU fallback: without a gfx950 device drt_create fails.
 */
#ifndef DARTRAY_RT_H
#define DARTRAY_RT_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DRT_OK 0
#define DRT_E_NODEVICE (-1) /* no HIP device / not gfx950 */
#define DRT_E_ARG (-2)      /* bad argument or call order */
#define DRT_E_HIP (-3)      /* HIP runtime error */
#define DRT_E_TRACE (-4)    /* the device path hit one of the reference's abort conditions
                               (negative length rt_routines.f90:1714, cell not found :1851/:2031) */
#define DRT_E_CAPACITY (-5) /* a ray queue outgrew its arena (ray_list.f90:310 grows instead) */
#define DRT_E_COMM (-6)     /* RCCL error */

/* rt_type ids, identical to rt_routines.f90:195-217 */
#define DRT_RTT_START 0
#define DRT_RTT_PRECALC_CELL 1
#define DRT_RTT_PRECALC_SRC 2
#define DRT_RTT_PREP_PART2 4
#define DRT_RTT_DIR_CELL 5
#define DRT_RTT_I_OBS_DIR_CELL 6
#define DRT_RTT_DIR_SRC 7
#define DRT_RTT_I_OBS_DIR_SRC 8
#define DRT_RTT_SCATT 10
#define DRT_RTT_I_OBS 11

/* Field selectors for drt_get_array / drt_set_array (module arrays of smooth_grid_routines.f90) */
#define DRT_A_U_FEST 1      /* u_fest_arr(0:lnum-1,0:tot_ncell-1)            :98  */
#define DRT_A_U_FINAL 2     /* u_final_arr(0:lnum-1,0:tot_ncell-1)           :102 */
#define DRT_A_SCASPE 3      /* scaspe_arr(il)%a(0:npix-1,0:tot_ncell-1)      :119 */
#define DRT_A_SCASPE_TOT 4  /* scaspe_tot_arr(il)%a                          :125 */
#define DRT_A_SCASPE_PREV 5 /* scaspe_prev_arr(il)%a                         :134 */
#define DRT_A_LUMCELL 6     /* lumcell(0:lnum-1,0:tot_ncell-1)               :188 */
#define DRT_A_LUM_LOST 7    /* lum_lost(0:lnum-1)             rt_routines.f90:131 */
#define DRT_A_I_OBS 8       /* i_obs_arr(0:lnum-1,0:tot_ncell+tot_p_src-1,0:tot_ndir-1) :140 */
#define DRT_A_DENS 9        /* dens_arr(0:lnum-1,0:tot_ncell-1)              :62  */
#define DRT_A_DENS_STARS 10 /* dens_stars_arr                                :70  */
#define DRT_A_TOT_RAD_EN 11 /* tot_rad_en(0:lnum-1)           rt_routines.f90:33  */
#define DRT_A_LUM_LOST_PREV 12 /* lum_lost_prev(0:lnum-1)     rt_routines.f90:133 */
#define DRT_A_PSEL_AV 14   /* [tot_ncell+tot_p_src] psel_av of every source of the last rt_loop, rt_routines.f90:1456-1465 (get only;
                            * needs drt_set_psel_av) */
#define DRT_A_PSEL_SUM 15  /* its numerator and denominator on the device (drt_device_ptr: summed over the ranks in rt_mpi) */
#define DRT_A_PSEL_CNT 16
#define DRT_A_SRC_COST 13  /* [tot_ncell] no reference counterpart: wave cycles spent on each cell source in the last cell-source
                            * rt_loop (zero elsewhere). Steers the task order / rank shares of the next one; with np_mpi > 1 it
                            * is summed over the ranks in rt_mpi like the fields (every source is traced by exactly one rank).
                            * drt_device_ptr hands out 2 * tot_ncell values: behind the costs lie the sources' queue lengths (the
                            * longest ray queue among a source's tasks, scaled to 48 sectors per task; the next launch cuts its
                            * tasks by them), to be summed over the ranks with them. */

/* rt_algorithm_ID (rt_routines.f90:232-250) as far as it changes the path: the stellar-emission run, and the i_obs algorithm
 * (dartray_i_obs, dartray_hub.f90:125-168: observer passes for new directions from a stored scaspe_tot) */
#define DRT_RTA_MAIN 0
#define DRT_RTA_I_OBS 1

typedef struct drt_ctx drt_ctx;

/* Scalars of rt_routines.f90 / smooth_grid_routines.f90 that steer the path. */
typedef struct drt_params {
    int32_t kp_sca_max;         /* rt_routines.f90:9   */
    int32_t bm_par;             /* :14 (INTEGER in the reference) */
    int32_t bm_par_sca;         /* :16 */
    int32_t bm_par_max;         /* :18 */
    double rad_lim;             /* :24 */
    double accuracy;            /* :31 -> en_lim by set_en_lim :3137 */
    double conv_en_lim;         /* :27 */
    double cs;                  /* light speed in csize units per s, set_units :5190 */
    double dist_obs;            /* :476 */
    int32_t wall_on[3];         /* x_wall_on, y_wall_on, z_wall_on  :497-501 */
    double wall_coord[6];       /* x/y/z_wall_coord(2) in model units (after set_walls :2600) */
    int32_t max_sca_iterations; /* smooth_grid_routines.f90:553; < 0 = unlimited */
    int32_t np_mpi;             /* number of ranks sharing the sources, rt_routines.f90:1003-1005 */
    int32_t id_mpi;             /* this rank */
} drt_params;

typedef struct drt_counters {
    int64_t intersections; /* passes through the main: loop body of ray_tracing, rt_routines.f90:1689 */
    int64_t rays;          /* calls of ray_tracing */
    int64_t deposits;      /* intersections with length /= 0 (:1731) */
    int64_t scaspe_adds;   /* (intersection, wavelength) pairs that issued a scaspe axpy (:2656) */
    int64_t lambda_deposits; /* (deposit, active wavelength) pairs: one dens_arr read each (:2380) */
    int64_t u_adds;        /* u_fest / u_final read-modify-writes issued (:2445, :2499) */
    int64_t u_reads;       /* u_fest reads of the blocked branch (:2477) */
    int64_t scaspe_slots;  /* scaspe elements read-modify-written (sum of npix over scaspe_adds) */
    int64_t sources;       /* sources processed by the last rt_loop */
    double kernel_ms;      /* device time of the ray-tracing kernel(s) of the last rt_loop (hipEvent), attempts that overflowed
                              and were undone included (`retries`) */
    int64_t launches;      /* kernel launches behind kernel_ms */
    int64_t atomic_instructions; /* f64 atomic wave-instructions issued for scaspe / u_final by the deposit loop */
    int64_t atomic_lanes;  /* active lanes summed over those instructions (8 bytes of payload each) */
    int64_t heavy_tasks;   /* last rt_loop: tasks of cell sources cut into fewer sectors per task (measured-cost scheduling) */
    int64_t deep_sources;  /* last rt_loop: cell sources run as sub-tree tasks */
    int64_t retries;       /* last rt_loop: attempts undone after a ray queue outgrew its arena slice and repeated with deeper queues */
} drt_counters;

/* ---- life cycle ------------------------------------------------------------------------------- */
int drt_create(drt_ctx **out, int device);
int drt_destroy(drt_ctx *ctx);
const char *drt_last_error(const drt_ctx *ctx);
/* arena_bytes: device memory for the per-wave ray queues; waves: persistent wavefronts (0 = as many as the chip keeps resident).
 * arena_bytes 0 = leave as it is; a context that was never given one sizes it when it is first needed: DRT_ARENA_GB from the
 * environment if set, else 16 GiB x max(1, (8 LP + 40) / 72) with LP = the wavelengths per ray rounded up to a power of two - the
 * queue DEPTH of 16 GiB at four wavelengths, e.g. 66 GiB at 32 - and at most a quarter of the device's memory and 60 % of what is free. Queues that still
 * overflow are handled (the launch is undone and repeated with finer tasks); a larger arena saves those repeats. */
int drt_configure(drt_ctx *ctx, size_t arena_bytes, int waves);

/* ---- module state in (grid_initialize, dartray_hub.f90:472-499) ----------------------------------- */
/* read_main_grid (io_routines.f90:297): cchild, lvl (int32), cindex (int64), ccoord(3,N), csize(N) */
int drt_set_grid(drt_ctx *ctx, int32_t tot_ncell, int32_t max_lvl, const int32_t base[2],
                 const int32_t *cchild, const int32_t *lvl, const int64_t *cindex,
                 const double *ccoord, const double *csize);
int drt_set_params(drt_ctx *ctx, const drt_params *p);
/* lnum and the per-wavelength opacities ksca_arr_norm, gsca_arr (smooth_grid_routines.f90:220-222) */
int drt_set_wavelengths(drt_ctx *ctx, int32_t lnum, const double *ksca_arr_norm, const double *gsca_arr);
/* read_p_src (io_routines.f90:620-666): ccoord_p_src(3,nsrc), lum_p_src_arr(lnum,nsrc) */
int drt_set_point_sources(drt_ctx *ctx, int32_t tot_p_src, const double *ccoord_p_src, const double *lum_p_src_arr);
/* read_dir_out (io_routines.f90:520-570): dir_i_out(tot_ndir,2) */
/* rta_main (default) or rta_i_obs: in the latter rt_loop_iobs(rtt_i_obs) takes the scattered luminosity towards an observer from
 * the HEALPix pixel of scaspe_tot that holds the direction (assign_src_lum, rt_routines.f90:1538-1548, 3043-3071); scaspe_tot is
 * handed over with drt_set_array(DRT_A_SCASPE_TOT) after drt_rt_prepare(rtt_prep_part2). */
int drt_set_algorithm(drt_ctx *ctx, int32_t rta);
/* print_psel_av (rt_routines.f90:269, 1456-1465): the average path of the rays of each source that ended in a stage, read with
 * drt_get_array(DRT_A_PSEL_AV) after the stage's rt_loop (and rt_mpi with several ranks) */
int drt_set_psel_av(drt_ctx *ctx, int32_t on);
int drt_set_observers(drt_ctx *ctx, int32_t tot_ndir, const double *dir_i_out);

/* ---- the three entry points + rt_prepare ---------------------------------------------------------- */
/* rt_prepare (dartray_hub.f90:386-469). rt_type = DRT_RTT_START: set_en_lim, prepare_p_src,
 * calc_total_luminosity (needs grid, params, wavelengths, dens, dens_stars); DRT_RTT_PREP_PART2:
 * create_scaspe, create_i_obs; DRT_RTT_SCATT: the scattering-iteration block (:432-457), after
 * which drt_cnflag() tells whether to stop; DRT_RTT_I_OBS: calc_total_luminosity_sca. */
int drt_rt_prepare(drt_ctx *ctx, int32_t rt_type);
/* rt_loop (rt_routines.f90:985): rt_type in {1,2,5,7,10} */
int drt_rt_loop(drt_ctx *ctx, int32_t rt_type);
/* rt_loop_iobs (rt_routines.f90:1470): rt_type in {6,8,11} */
int drt_rt_loop_iobs(drt_ctx *ctx, int32_t rt_type);
/* rt_mpi (io_routines.f90:3667): stage-end reductions for rt_type in {2,7,10}; all-reduce over the
 * communicator attached with drt_comm_init (identity + the u_final clamp on one rank) */
int drt_rt_mpi(drt_ctx *ctx, int32_t rt_type);
/* For a host that performs rt_mpi's reductions itself on the device buffers (drt_device_ptr): after it has also summed
 * DRT_A_SRC_COST over the ranks it says so here, and the next cell-source stage deals its sources among the ranks by
 * those costs. Without the call the deal uses a function of lumcell alone (identical on all ranks by construction). */
int drt_mark_cost_reduced(drt_ctx *ctx);
/* The deal itself, a pure host function (no context, no GPU): owner[k] = rank of source k of n consecutive emitting cells with
 * expected costs cost[k], np ranks, chunks of at most `chunk` sources. Exported for tests of the multi-rank logic. */
int drt_deal_by_cost(const double *cost, int64_t n, int32_t np, int32_t chunk, int32_t *owner);

/* ---- module state out -------------------------------------------------------------------------------- */
/* Copy a field to/from the host. il selects the wavelength for the scaspe arrays (ignored
 * otherwise). count must equal the field's element count. */
int drt_get_array(drt_ctx *ctx, int32_t which, int32_t il, double *host, size_t count);
int drt_set_array(drt_ctx *ctx, int32_t which, int32_t il, const double *host, size_t count);
/* Device address and element count of a field, for hosts that run the stage-end reductions
 * themselves (e.g. torch.distributed / RCCL on the same buffers). The memory stays owned by ctx.
 * The scaspe arrays are held with padded rows on the device (row length = npix rounded up to a multiple of 8, so that
 * every cell's row starts on a 64-byte line); count is that padded size and the padding is zero, so element-wise sums
 * over ranks are unaffected. drt_get_array / drt_set_array exchange the dense (npix, tot_ncell) arrays of the reference. */
int drt_device_ptr(drt_ctx *ctx, int32_t which, int32_t il, void **dev_ptr, size_t *count);
/* npix_arr, npix_hp_arr, kp_sca_arr of wavelength il (set_npix_arr, rt_routines.f90:5235) */
int drt_get_npix(const drt_ctx *ctx, int32_t il, int32_t *npix, int32_t *npix_hp, int32_t *kp);
int drt_get_src_cell(const drt_ctx *ctx, int32_t *src_cell /* [tot_ncell] */, double *ccoord_p_src /* (3,nsrc) or NULL */);
double drt_en_lim(const drt_ctx *ctx);
int32_t drt_iterations(const drt_ctx *ctx);
int32_t drt_cnflag(const drt_ctx *ctx);
/* calc_sed + sum_i_obs for units 'erg/s/Hz/pc^2/sr' (sed_routines.f90:90-146, 211-246): sed(lnum,tot_ndir) */
int drt_calc_sed(drt_ctx *ctx, double *sed);
int drt_get_counters(const drt_ctx *ctx, drt_counters *out);
/* The cells THIS rank traces in drt_rt_loop / drt_rt_loop_iobs(rt_type) - the loop `i = im*np_mpi + id_mpi` of
 * rt_routines.f90:1003-1016 after its skips (parents, zero luminosity). Cell sources are shared among the np_mpi ranks in
 * chunks of at most 64 consecutive emitting cells (DRT_PARTITION_CHUNK), dealt by EXPECTED COST, heaviest chunk to the
 * least loaded rank (longest-processing-time-first; drt_deal_by_cost is that function). The expected cost of a source is
 * its measured wave time in the previous launch of the same kind of stage (DRT_A_SRC_COST) scaled with its luminosity -
 * used only once those times have been summed over the ranks (drt_rt_mpi, or drt_mark_cost_reduced after the host's own
 * reduction) - and sqrt(luminosity) otherwise, so the deal is the same on every rank. The environment variable
 * DRT_PARTITION selects another deal: `mod` = the reference's every-np-th cell, `block` = one contiguous block per rank,
 * `cyclic` = the chunks dealt round the ranks. Point sources are listed on every rank, which then runs every np_mpi-th
 * (source, sector) task of them (rt_routines.f90:1356-1358).
 * The list returned is the one a drt_rt_loop(rt_type) called NOW would trace: it holds for the current state of the costs
 * (it changes when drt_rt_mpi / drt_mark_cost_reduced mark them as summed, and after every launch). Side effects: the
 * query uploads the list to the device source list and recomputes the expected costs, as drt_rt_loop itself does before
 * a launch; nothing a following drt_rt_loop relies on. */
int drt_get_sources(drt_ctx *ctx, int32_t rt_type, int32_t *cells, int32_t capacity, int32_t *count);
int drt_reset_counters(drt_ctx *ctx);
/* Introspection (no GPU needed): how a cell-source launch of `n` sources with expected costs `cost` is cut into tasks for `waves`
 * persistent wavefronts at `G` sectors per task (see plan_tasks in dartray_rt.hip): cls[k] = 0 light (G sectors per task), -1 deep
 * (sub-tree tasks), else the sectors per task of a heavy source; returns the number of explicit heavy tasks, *limit = the largest
 * cost a task may have. measured: the costs are measured wave times (else the prior); f_last / q_fill / G_last: task limit,
 * queue fill and sectors of the stage's last launch (0: none); q48 (may be NULL) / cap: the sources' own queue lengths at 48
 * sectors per task (0: not known) and the rays a queue holds. */
int drt_plan_tasks(const double *cost, int64_t n, int32_t waves, int32_t G, int32_t nbatch, double heavy_div, double deep_min,
                   int32_t measured, double f_last, double q_fill, int32_t G_last, const double *q48, double cap, int32_t *cls,
                   double *limit);

/* Introspection (no GPU needed): the line log q = a + b log lum through the sources whose longest ray queue q (at 48 sectors per
 * task) an undone launch measured, by which the queues of the others are estimated (fit_queue_prior in dartray_rt.hip). DRT_OK, or
 * DRT_E_CAPACITY when fewer than 64 pairs with q > 0 and lum > 0 exist or the luminosities are all alike. */
int drt_fit_queue_prior(const double *lum, const double *q, int64_t n, double *a, double *b);
/* Introspection (no GPU needed): how an arena of `arena_bytes` is split among `waves` wavefronts carrying `lp`
 * (a power of two >= lnum) luminosities per ray: queue depth `cap`, hash size `hcap`, bytes per wave `stride`, and the
 * byte offsets of the regions of one wave's slice: 4 ray queues, hash keys, hash members, status words, HG
 * normalisations, intensities, end. DRT_E_CAPACITY when fewer than 64 rays per queue would fit. */
int drt_arena_carve(size_t arena_bytes, int32_t waves, int32_t lp, int32_t *cap, int32_t *hcap, uint64_t *stride,
                    uint64_t offsets[10]);

/* ---- multi-GPU (one process per GPU; rt_mpi's MPI_Allreduce becomes ncclAllReduce over xGMI) ---- */
#define DRT_UNIQUE_ID_BYTES 128
int drt_comm_unique_id(char id[DRT_UNIQUE_ID_BYTES]);
int drt_comm_init(drt_ctx *ctx, int32_t nranks, int32_t rank, const char id[DRT_UNIQUE_ID_BYTES]);
int drt_comm_destroy(drt_ctx *ctx);
/* The RCCL path of this library end to end on one rank (find the library, one-rank communicator from a fresh id,
 * ncclAllReduce(double, sum) in place on a device buffer, destroy): DRT_OK, or DRT_E_COMM with drt_last_error(). */
int drt_comm_selftest(drt_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
