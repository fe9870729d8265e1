// dartray_io.cpp - HDF5 grid input / field output in the reference's on-disk layout (include/dartray_io.h).
// Host code only; HDF5 C API 1.10. Every function cites the io_routines.f90 routine it restates.
#include "../../include/dartray_io.h"

#include <hdf5.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {
thread_local std::string g_err;
int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
struct Quiet {  // the reference tests `error` itself; keep the HDF5 error stack off stderr
    Quiet() { H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr); }
};
struct File {
    hid_t id = -1;
    ~File() { if (id >= 0) H5Fclose(id); }
};
// Fortran dims (d1, d2, ...) -> C dims reversed (what h5screate_simple_f does)
std::vector<hsize_t> cdims(int rank, const int64_t* fdims)
{
    std::vector<hsize_t> d((size_t)rank);
    for (int i = 0; i < rank; i++) d[(size_t)i] = (hsize_t)fdims[rank - 1 - i];
    return d;
}
int write_dset(hid_t file, const char* name, hid_t ftype, hid_t mtype, int rank, const int64_t* fdims, const void* data)
{
    std::vector<hsize_t> d = cdims(rank, fdims);
    hid_t sp = H5Screate_simple(rank, d.data(), nullptr);
    if (sp < 0) return fail(DRT_IO_E_WRITE, "dataspace for '%s'", name);
    hid_t ds = H5Dcreate2(file, name, ftype, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    if (ds < 0) { H5Sclose(sp); return fail(DRT_IO_E_DATASET, "cannot create dataset '%s'", name); }
    herr_t e = H5Dwrite(ds, mtype, H5S_ALL, H5S_ALL, H5P_DEFAULT, data);
    H5Dclose(ds);
    H5Sclose(sp);
    if (e < 0) return fail(DRT_IO_E_WRITE, "write of dataset '%s' failed", name);
    return DRT_IO_OK;
}
// reads a dataset after checking its extent against the expected (Fortran-order) dims, like read_lambda_grid :437-444
int read_dset(hid_t file, const char* name, hid_t mtype, int rank, const int64_t* fdims, void* data)
{
    hid_t ds = H5Dopen2(file, name, H5P_DEFAULT);
    if (ds < 0) return fail(DRT_IO_E_DATASET, "dataset '%s' not found", name);
    hid_t sp = H5Dget_space(ds);
    int r = H5Sget_simple_extent_ndims(sp);
    std::vector<hsize_t> got((size_t)(r > 0 ? r : 1));
    if (r > 0) H5Sget_simple_extent_dims(sp, got.data(), nullptr);
    H5Sclose(sp);
    std::vector<hsize_t> want = cdims(rank, fdims);
    bool ok = (r == rank);
    for (int i = 0; ok && i < rank; i++) ok = (got[(size_t)i] == want[(size_t)i]);
    if (!ok) { H5Dclose(ds); return fail(DRT_IO_E_DIMS, "dataset '%s': dimensions do not coincide with the expected ones", name); }
    herr_t e = H5Dread(ds, mtype, H5S_ALL, H5S_ALL, H5P_DEFAULT, data);
    H5Dclose(ds);
    if (e < 0) return fail(DRT_IO_E_READ, "read of dataset '%s' failed", name);
    return DRT_IO_OK;
}
// write(label,'(F9.3)') lambda ; trim(adjustl(label))  (io_routines.f90:2301)
std::string wave_label(double lambda)
{
    char buf[64];
    snprintf(buf, sizeof buf, "%9.3f", lambda);
    std::string s(buf);
    if (s.size() > 9) s = std::string(9, '*');  // Fortran fills the field with asterisks on overflow
    size_t b = s.find_first_not_of(' ');
    return b == std::string::npos ? std::string() : s.substr(b);
}
}  // namespace

extern "C" const char* drt_io_last_error(void) { return g_err.c_str(); }

// print_3d_grid_file (io_routines.f90:202-294)
extern "C" int drt_io_write_main_grid(const char* path, int32_t N, const int32_t* ncell, const double* ccoord, const int32_t* cchild,
                                      const int64_t* cindex, const int32_t* lvl, const double* csize, const double* dens,
                                      const double* dens_stars, const int32_t base[2])
{
    if (!path || N <= 0 || !ncell || !ccoord || !cchild || !cindex || !lvl || !csize || !dens || !dens_stars || !base)
        return fail(DRT_IO_E_ARG, "drt_io_write_main_grid: null argument");
    Quiet q;
    File f;
    f.id = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    if (f.id < 0) return fail(DRT_IO_E_OPEN, "cannot create '%s'", path);
    const int64_t d1[1] = {N}, d2[2] = {3, N}, db[1] = {2};
    int rc;
    // order and types of make_dsetname_main_grid (:184-198) and the select case at :233-243
    if ((rc = write_dset(f.id, "ncell", H5T_NATIVE_INT32, H5T_NATIVE_INT32, 1, d1, ncell))) return rc;
    if ((rc = write_dset(f.id, "ccoord", H5T_NATIVE_DOUBLE, H5T_NATIVE_DOUBLE, 2, d2, ccoord))) return rc;
    if ((rc = write_dset(f.id, "cchild", H5T_NATIVE_INT32, H5T_NATIVE_INT32, 1, d1, cchild))) return rc;
    if ((rc = write_dset(f.id, "cindex", H5T_NATIVE_INT64, H5T_NATIVE_INT64, 1, d1, cindex))) return rc;
    if ((rc = write_dset(f.id, "lvl", H5T_NATIVE_INT32, H5T_NATIVE_INT32, 1, d1, lvl))) return rc;
    if ((rc = write_dset(f.id, "csize", H5T_NATIVE_DOUBLE, H5T_NATIVE_DOUBLE, 1, d1, csize))) return rc;
    if ((rc = write_dset(f.id, "dens", H5T_NATIVE_DOUBLE, H5T_NATIVE_DOUBLE, 1, d1, dens))) return rc;
    if ((rc = write_dset(f.id, "dens_stars", H5T_NATIVE_DOUBLE, H5T_NATIVE_DOUBLE, 1, d1, dens_stars))) return rc;
    if ((rc = write_dset(f.id, "base", H5T_NATIVE_INT32, H5T_NATIVE_INT32, 1, db, base))) return rc;
    return DRT_IO_OK;
}

// read_main_grid (:319-333): tot_ncell is the number of points of dataset 'ncell'
extern "C" int drt_io_main_grid_size(const char* path, int32_t* tot_ncell)
{
    if (!path || !tot_ncell) return fail(DRT_IO_E_ARG, "drt_io_main_grid_size: null argument");
    Quiet q;
    File f;
    f.id = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
    if (f.id < 0) return fail(DRT_IO_E_OPEN, "main grid file not found: '%s'", path);
    hid_t ds = H5Dopen2(f.id, "ncell", H5P_DEFAULT);
    if (ds < 0) return fail(DRT_IO_E_DATASET, "dataset 'ncell' not found");
    hid_t sp = H5Dget_space(ds);
    hssize_t n = H5Sget_simple_extent_npoints(sp);
    H5Sclose(sp);
    H5Dclose(ds);
    if (n <= 0 || n > INT32_MAX) return fail(DRT_IO_E_DIMS, "dataset 'ncell' has %lld points", (long long)n);
    *tot_ncell = (int32_t)n;
    return DRT_IO_OK;
}

// read_main_grid (:297-394)
extern "C" int drt_io_read_main_grid(const char* path, int32_t N, int32_t* ncell, double* ccoord, int32_t* cchild, int64_t* cindex,
                                     int32_t* lvl, double* csize, double* dens_ref, double* dens_stars_ref, int32_t base[2])
{
    if (!path || N <= 0 || !ncell || !ccoord || !cchild || !cindex || !lvl || !csize || !dens_ref || !dens_stars_ref || !base)
        return fail(DRT_IO_E_ARG, "drt_io_read_main_grid: null argument");
    Quiet q;
    File f;
    f.id = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
    if (f.id < 0) return fail(DRT_IO_E_OPEN, "main grid file not found: '%s'", path);
    const int64_t d1[1] = {N}, d2[2] = {3, N}, db[1] = {2};
    int rc;
    if ((rc = read_dset(f.id, "ncell", H5T_NATIVE_INT32, 1, d1, ncell))) return rc;
    if ((rc = read_dset(f.id, "ccoord", H5T_NATIVE_DOUBLE, 2, d2, ccoord))) return rc;
    if ((rc = read_dset(f.id, "cchild", H5T_NATIVE_INT32, 1, d1, cchild))) return rc;
    if ((rc = read_dset(f.id, "cindex", H5T_NATIVE_INT64, 1, d1, cindex))) return rc;
    if ((rc = read_dset(f.id, "lvl", H5T_NATIVE_INT32, 1, d1, lvl))) return rc;
    if ((rc = read_dset(f.id, "csize", H5T_NATIVE_DOUBLE, 1, d1, csize))) return rc;
    if ((rc = read_dset(f.id, "dens", H5T_NATIVE_DOUBLE, 1, d1, dens_ref))) return rc;
    if ((rc = read_dset(f.id, "dens_stars", H5T_NATIVE_DOUBLE, 1, d1, dens_stars_ref))) return rc;
    if ((rc = read_dset(f.id, "base", H5T_NATIVE_INT32, 1, db, base))) return rc;
    return DRT_IO_OK;
}

// the lambda grids read by read_lambda_grid (:397-470): datasets dens, dens_stars of tot_ncell elements
extern "C" int drt_io_write_lambda_grid(const char* path, int32_t N, const double* dens, const double* dens_stars)
{
    if (!path || N <= 0 || !dens || !dens_stars) return fail(DRT_IO_E_ARG, "drt_io_write_lambda_grid: null argument");
    Quiet q;
    File f;
    f.id = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    if (f.id < 0) return fail(DRT_IO_E_OPEN, "cannot create '%s'", path);
    const int64_t d1[1] = {N};
    int rc;
    if ((rc = write_dset(f.id, "dens", H5T_NATIVE_DOUBLE, H5T_NATIVE_DOUBLE, 1, d1, dens))) return rc;
    if ((rc = write_dset(f.id, "dens_stars", H5T_NATIVE_DOUBLE, H5T_NATIVE_DOUBLE, 1, d1, dens_stars))) return rc;
    return DRT_IO_OK;
}

extern "C" int drt_io_read_lambda_grid(const char* path, int32_t N, double* dens, double* dens_stars)
{
    if (!path || N <= 0 || !dens || !dens_stars) return fail(DRT_IO_E_ARG, "drt_io_read_lambda_grid: null argument");
    Quiet q;
    File f;
    f.id = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
    if (f.id < 0) return fail(DRT_IO_E_OPEN, "lambda grid file not found: '%s'", path);
    const int64_t d1[1] = {N};
    int rc;
    if ((rc = read_dset(f.id, "dens", H5T_NATIVE_DOUBLE, 1, d1, dens))) return rc;
    if ((rc = read_dset(f.id, "dens_stars", H5T_NATIVE_DOUBLE, 1, d1, dens_stars))) return rc;
    return DRT_IO_OK;
}

// print_big_array (:1112-1178): a new file with ONE double dataset
extern "C" int drt_io_write_array(const char* path, const char* dsetname, int rank, const int64_t* fdims, const double* data)
{
    if (!path || !dsetname || rank < 1 || rank > 4 || !fdims || !data) return fail(DRT_IO_E_ARG, "drt_io_write_array: bad argument");
    Quiet q;
    File f;
    f.id = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    if (f.id < 0) return fail(DRT_IO_E_OPEN, "cannot create '%s'", path);
    return write_dset(f.id, dsetname, H5T_NATIVE_DOUBLE, H5T_NATIVE_DOUBLE, rank, fdims, data);
}

// read_big_array (the counterpart used by read_output :1900-1997 for restarts)
extern "C" int drt_io_read_array(const char* path, const char* dsetname, int rank, const int64_t* fdims, double* data)
{
    if (!path || !dsetname || rank < 1 || rank > 4 || !fdims || !data) return fail(DRT_IO_E_ARG, "drt_io_read_array: bad argument");
    Quiet q;
    File f;
    f.id = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
    if (f.id < 0) return fail(DRT_IO_E_OPEN, "file not found: '%s'", path);
    return read_dset(f.id, dsetname, H5T_NATIVE_DOUBLE, rank, fdims, data);
}

// extent of a dataset in Fortran order (find_out_tot_ndir_scaspe :2566-2651 asks the scaspe_tot files of an earlier run)
extern "C" int drt_io_array_dims(const char* path, const char* dsetname, int rank, int64_t* fdims)
{
    if (!path || !dsetname || rank < 1 || rank > 4 || !fdims) return fail(DRT_IO_E_ARG, "drt_io_array_dims: bad argument");
    Quiet q;
    File f;
    f.id = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
    if (f.id < 0) return fail(DRT_IO_E_OPEN, "file not found: '%s'", path);
    hid_t ds = H5Dopen2(f.id, dsetname, H5P_DEFAULT);
    if (ds < 0) return fail(DRT_IO_E_DATASET, "dataset '%s' not found", dsetname);
    hid_t sp = H5Dget_space(ds);
    const int r = H5Sget_simple_extent_ndims(sp);
    hsize_t got[8] = {0};
    if (r == rank) H5Sget_simple_extent_dims(sp, got, nullptr);
    H5Sclose(sp);
    H5Dclose(ds);
    if (r != rank) return fail(DRT_IO_E_DATASET, "dataset '%s' has rank %d, expected %d", dsetname, r, rank);
    for (int i = 0; i < rank; i++) fdims[i] = (int64_t)got[rank - 1 - i];
    return DRT_IO_OK;
}

// add_info_big_array (:1180-1240) for the integer 'nside_sca'
extern "C" int drt_io_add_int(const char* path, const char* dsetname, int32_t value)
{
    if (!path || !dsetname) return fail(DRT_IO_E_ARG, "drt_io_add_int: null argument");
    Quiet q;
    File f;
    f.id = H5Fopen(path, H5F_ACC_RDWR, H5P_DEFAULT);
    if (f.id < 0) return fail(DRT_IO_E_OPEN, "file not found: '%s'", path);
    const int64_t d1[1] = {1};
    return write_dset(f.id, dsetname, H5T_NATIVE_INT32, H5T_NATIVE_INT32, 1, d1, &value);
}

extern "C" int drt_io_read_int(const char* path, const char* dsetname, int32_t* value)
{
    if (!path || !dsetname || !value) return fail(DRT_IO_E_ARG, "drt_io_read_int: null argument");
    Quiet q;
    File f;
    f.id = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
    if (f.id < 0) return fail(DRT_IO_E_OPEN, "file not found: '%s'", path);
    const int64_t d1[1] = {1};
    return read_dset(f.id, dsetname, H5T_NATIVE_INT32, 1, d1, value);
}

// set_filenames (:757-790)
extern "C" int drt_io_filename(char* out, size_t cap, const char* label, double lambda, const char* what, const char* chext)
{
    if (!out || !label || !what) return fail(DRT_IO_E_ARG, "drt_io_filename: null argument");
    std::string s = "grid_" + std::string(label);
    if (lambda >= 0) s += "_l" + wave_label(lambda) + "um";
    s += "_" + std::string(what) + std::string(chext ? chext : "") + ".h5";
    if (s.size() + 1 > cap) return fail(DRT_IO_E_ARG, "drt_io_filename: buffer too small");
    memcpy(out, s.c_str(), s.size() + 1);
    return DRT_IO_OK;
}

// grid_file_lambda_arr (:2327)
extern "C" int drt_io_lambda_grid_filename(char* out, size_t cap, const char* label, double lambda)
{
    if (!out || !label) return fail(DRT_IO_E_ARG, "drt_io_lambda_grid_filename: null argument");
    std::string s = "grid_" + std::string(label) + "_l" + wave_label(lambda) + "um.h5";
    if (s.size() + 1 > cap) return fail(DRT_IO_E_ARG, "drt_io_lambda_grid_filename: buffer too small");
    memcpy(out, s.c_str(), s.size() + 1);
    return DRT_IO_OK;
}
