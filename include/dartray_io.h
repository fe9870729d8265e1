/* dartray_io.h - HDF5 grid input and field output of DART-Ray, C ABI (host only, no GPU code).
 *
 * the on-disk contract of the reference, so that grids written by the
 * reference's builders can be fed to the GPU path and its output files read by the reference's tools
 * (and the other way round). The reference does this through the HDF5 Fortran API in io_routines.f90;
 * this library uses the HDF5 C API. Arrays are passed exactly as the Fortran module variables lie in
 * memory (column-major); dimension lists are given in FORTRAN order and written reversed, which is
 * what the HDF5 Fortran wrappers do, so files are interchangeable byte for byte at the dataset level.
 *
 *   reference routine (io_routines.f90)                      entry point
 *   print_3d_grid_file :202-294, make_dsetname_main_grid :184  drt_io_write_main_grid
 *   read_main_grid :297-394                                    drt_io_main_grid_size, drt_io_read_main_grid
 *   read_lambda_grid :397-470 (datasets dens, dens_stars)      drt_io_read_lambda_grid / drt_io_write_lambda_grid
 *   print_big_array :1112-1178, read_big_array                 drt_io_write_array / drt_io_read_array
 *   add_info_big_array :1180-1240 (nside_sca)                  drt_io_add_int / drt_io_read_int
 *   set_filenames :720-794, label_wave_arr :2301 (F9.3)        drt_io_filename, drt_io_lambda_grid_filename
 *
 * All functions return 0 on success, a negative DRT_IO_E_* code otherwise; drt_io_last_error() gives text.
 */
#ifndef DARTRAY_IO_H
#define DARTRAY_IO_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { DRT_IO_OK = 0, DRT_IO_E_OPEN = -1, DRT_IO_E_DATASET = -2, DRT_IO_E_DIMS = -3, DRT_IO_E_WRITE = -4, DRT_IO_E_READ = -5, DRT_IO_E_ARG = -6 };

const char* drt_io_last_error(void);

/* Main grid file: datasets ncell, cchild, lvl (int32 [N]), cindex (int64 [N]), ccoord (f64, Fortran (3,N) -> file
 * dims [N][3]), csize, dens, dens_stars (f64 [N]), base (int32 [2]). */
int drt_io_write_main_grid(const char* path, int32_t tot_ncell, const int32_t* ncell, const double* ccoord,
                           const int32_t* cchild, const int64_t* cindex, const int32_t* lvl, const double* csize,
                           const double* dens, const double* dens_stars, const int32_t base[2]);
/* tot_ncell = number of elements of dataset 'ncell' (read_main_grid :330-333) */
int drt_io_main_grid_size(const char* path, int32_t* tot_ncell);
int drt_io_read_main_grid(const char* path, int32_t tot_ncell, int32_t* ncell, double* ccoord, int32_t* cchild,
                          int64_t* cindex, int32_t* lvl, double* csize, double* dens_ref, double* dens_stars_ref,
                          int32_t base[2]);

/* Per-wavelength grid file: datasets dens and dens_stars, f64 [N]; the size is checked like the reference does. */
int drt_io_write_lambda_grid(const char* path, int32_t tot_ncell, const double* dens, const double* dens_stars);
int drt_io_read_lambda_grid(const char* path, int32_t tot_ncell, double* dens, double* dens_stars);

/* One f64 dataset per file (print_big_array): u_fest, u_final [N]; scaspe, scaspe_tot (npix, N); i_obs
 * (N + nsrc, ndir); sed_arr, sed_arr_dir (lnum, ndir); lum_lost [lnum]; psel_av_arr. `fdims` in Fortran order. */
int drt_io_write_array(const char* path, const char* dsetname, int rank, const int64_t* fdims, const double* data);
int drt_io_read_array(const char* path, const char* dsetname, int rank, const int64_t* fdims, double* data);
/* extent of a dataset, Fortran order (find_out_tot_ndir_scaspe, io_routines.f90:2566) */
int drt_io_array_dims(const char* path, const char* dsetname, int rank, int64_t* fdims);
/* extra int32 [1] dataset appended to an existing file (nside_sca of the scaspe_tot files) */
int drt_io_add_int(const char* path, const char* dsetname, int32_t value);
int drt_io_read_int(const char* path, const char* dsetname, int32_t* value);

/* 'grid_<label>_l<lambda F9.3, left-adjusted>um_<what><chext>.h5' for per-wavelength files
 * (what = u_fest_part1, ufield_part2, ufield, scaspe_part2, scaspe_tot, i_obs_part2, i_obs), and
 * 'grid_<label>_<what><chext>.h5' when lambda < 0 (what = sed, sed_dir, lum_lost, lum_lost_part2, psel_av ...). */
int drt_io_filename(char* out, size_t cap, const char* label_model_out, double lambda, const char* what, const char* chext);
/* 'grid_<label_model_lambda_grid>_l<lambda>um.h5' (io_routines.f90:2327) */
int drt_io_lambda_grid_filename(char* out, size_t cap, const char* label_model_lambda_grid, double lambda);

#ifdef __cplusplus
}
#endif
#endif
