!> Fortran host side of the MI355X ray-cell path: ISO_C_BINDING interfaces to libdartray_rt.so
!> (include/dartray_rt.h) and a facade that keeps the reference's driver API.
!>
!> The reference's hot path sits behind argument-less module procedures that communicate through
!> module variables: rt_loop (rt_routines.f90:985), rt_loop_iobs (:1470), rt_mpi
!> (io_routines.f90:3667), rt_prepare (dartray_hub.f90:386) and the state machine dartray_main
!> (dartray_hub.f90:12-67). The facade below has the same procedure names and reads/writes
!> variables with the same names, shapes and bounds as smooth_grid_routines.f90:42-222 and
!> rt_routines.f90:9-131. To drop it into the reference, replace `use dartray_state` by
!> `use smooth_grid_routines` / `use rt_routines` (see INTEGRATION.md).
module dartray_c
  use iso_c_binding
  implicit none

  integer(c_int), parameter :: DRT_A_U_FEST = 1, DRT_A_U_FINAL = 2, DRT_A_SCASPE = 3, DRT_A_SCASPE_TOT = 4, &
       DRT_A_SCASPE_PREV = 5, DRT_A_LUMCELL = 6, DRT_A_LUM_LOST = 7, DRT_A_I_OBS = 8, DRT_A_DENS = 9, &
       DRT_A_DENS_STARS = 10, DRT_A_TOT_RAD_EN = 11, DRT_A_PSEL_AV = 14

  type, bind(C) :: drt_params
     integer(c_int32_t) :: kp_sca_max, bm_par, bm_par_sca, bm_par_max
     real(c_double) :: rad_lim, accuracy, conv_en_lim, cs, dist_obs
     integer(c_int32_t) :: wall_on(3)
     real(c_double) :: wall_coord(6)
     integer(c_int32_t) :: max_sca_iterations, np_mpi, id_mpi
  end type drt_params

  type, bind(C) :: drt_counters
     integer(c_int64_t) :: intersections, rays, deposits, scaspe_adds, lambda_deposits, u_adds, u_reads, scaspe_slots, sources
     real(c_double) :: kernel_ms
     integer(c_int64_t) :: launches, atomic_instructions, atomic_lanes, heavy_tasks, deep_sources, retries
  end type drt_counters

  interface
     integer(c_int) function drt_create(ctx, device) bind(C, name='drt_create')
       import :: c_ptr, c_int
       type(c_ptr), intent(out) :: ctx
       integer(c_int), value :: device
     end function drt_create
     integer(c_int) function drt_destroy(ctx) bind(C, name='drt_destroy')
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx
     end function drt_destroy
     type(c_ptr) function drt_last_error(ctx) bind(C, name='drt_last_error')
       import :: c_ptr
       type(c_ptr), value :: ctx
     end function drt_last_error
     integer(c_int) function drt_configure(ctx, arena_bytes, waves) bind(C, name='drt_configure')
       import :: c_ptr, c_int, c_size_t
       type(c_ptr), value :: ctx
       integer(c_size_t), value :: arena_bytes
       integer(c_int), value :: waves
     end function drt_configure
     integer(c_int) function drt_set_grid(ctx, tot_ncell, max_lvl, base, cchild, lvl, cindex, ccoord, csize) bind(C, name='drt_set_grid')
       import :: c_ptr, c_int, c_int32_t, c_int64_t, c_double
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: tot_ncell, max_lvl
       integer(c_int32_t), intent(in) :: base(2), cchild(*), lvl(*)
       integer(c_int64_t), intent(in) :: cindex(*)
       real(c_double), intent(in) :: ccoord(3, *), csize(*)
     end function drt_set_grid
     integer(c_int) function drt_set_params(ctx, p) bind(C, name='drt_set_params')
       import :: c_ptr, c_int, drt_params
       type(c_ptr), value :: ctx
       type(drt_params), intent(in) :: p
     end function drt_set_params
     integer(c_int) function drt_set_wavelengths(ctx, lnum, ksca_arr_norm, gsca_arr) bind(C, name='drt_set_wavelengths')
       import :: c_ptr, c_int, c_int32_t, c_double
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: lnum
       real(c_double), intent(in) :: ksca_arr_norm(*), gsca_arr(*)
     end function drt_set_wavelengths
     integer(c_int) function drt_set_point_sources(ctx, tot_p_src, ccoord_p_src, lum_p_src_arr) bind(C, name='drt_set_point_sources')
       import :: c_ptr, c_int, c_int32_t, c_double
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: tot_p_src
       real(c_double), intent(in) :: ccoord_p_src(3, *), lum_p_src_arr(*)
     end function drt_set_point_sources
     integer(c_int) function drt_set_observers(ctx, tot_ndir, dir_i_out) bind(C, name='drt_set_observers')
       import :: c_ptr, c_int, c_int32_t, c_double
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: tot_ndir
       real(c_double), intent(in) :: dir_i_out(*)
     end function drt_set_observers
     integer(c_int) function drt_rt_prepare(ctx, rt_type) bind(C, name='drt_rt_prepare')
       import :: c_ptr, c_int, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: rt_type
     end function drt_rt_prepare
     integer(c_int) function drt_rt_loop(ctx, rt_type) bind(C, name='drt_rt_loop')
       import :: c_ptr, c_int, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: rt_type
     end function drt_rt_loop
     integer(c_int) function drt_rt_loop_iobs(ctx, rt_type) bind(C, name='drt_rt_loop_iobs')
       import :: c_ptr, c_int, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: rt_type
     end function drt_rt_loop_iobs
     integer(c_int) function drt_rt_mpi(ctx, rt_type) bind(C, name='drt_rt_mpi')
       import :: c_ptr, c_int, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: rt_type
     end function drt_rt_mpi
     integer(c_int) function drt_get_array(ctx, which, il, host, count) bind(C, name='drt_get_array')
       import :: c_ptr, c_int, c_int32_t, c_double, c_size_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: which, il
       real(c_double), intent(out) :: host(*)
       integer(c_size_t), value :: count
     end function drt_get_array
     integer(c_int) function drt_set_array(ctx, which, il, host, count) bind(C, name='drt_set_array')
       import :: c_ptr, c_int, c_int32_t, c_double, c_size_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: which, il
       real(c_double), intent(in) :: host(*)
       integer(c_size_t), value :: count
     end function drt_set_array
     integer(c_int) function drt_get_npix(ctx, il, npix, npix_hp, kp) bind(C, name='drt_get_npix')
       import :: c_ptr, c_int, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: il
       integer(c_int32_t), intent(out) :: npix, npix_hp, kp
     end function drt_get_npix
     integer(c_int) function drt_get_src_cell(ctx, src_cell, ccoord_p_src) bind(C, name='drt_get_src_cell')
       import :: c_ptr, c_int, c_int32_t, c_double
       type(c_ptr), value :: ctx
       integer(c_int32_t), intent(out) :: src_cell(*)
       real(c_double), intent(out) :: ccoord_p_src(3, *)
     end function drt_get_src_cell
     real(c_double) function drt_en_lim(ctx) bind(C, name='drt_en_lim')
       import :: c_ptr, c_double
       type(c_ptr), value :: ctx
     end function drt_en_lim
     integer(c_int32_t) function drt_iterations(ctx) bind(C, name='drt_iterations')
       import :: c_ptr, c_int32_t
       type(c_ptr), value :: ctx
     end function drt_iterations
     integer(c_int32_t) function drt_cnflag(ctx) bind(C, name='drt_cnflag')
       import :: c_ptr, c_int32_t
       type(c_ptr), value :: ctx
     end function drt_cnflag
     integer(c_int) function drt_calc_sed(ctx, sed) bind(C, name='drt_calc_sed')
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: sed(*)
     end function drt_calc_sed
     integer(c_int) function drt_get_counters(ctx, cnt) bind(C, name='drt_get_counters')
       import :: c_ptr, c_int, drt_counters
       type(c_ptr), value :: ctx
       type(drt_counters), intent(out) :: cnt
     end function drt_get_counters
     integer(c_int) function drt_get_sources(ctx, rt_type, cells, capacity, count) bind(C, name='drt_get_sources')
       import :: c_ptr, c_int, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: rt_type, capacity
       integer(c_int32_t), intent(out) :: cells(*), count
     end function drt_get_sources
     ! multi-GPU: one rank per GPU, rt_mpi's MPI_Allreduce becomes ncclAllReduce (RCCL over xGMI) on the device buffers
     integer(c_int) function drt_set_psel_av(ctx, on) bind(C, name='drt_set_psel_av')
       import :: c_ptr, c_int, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: on
     end function drt_set_psel_av
     integer(c_int) function drt_set_algorithm(ctx, rta) bind(C, name='drt_set_algorithm')
       import :: c_ptr, c_int, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: rta
     end function drt_set_algorithm
     integer(c_int) function drt_mark_cost_reduced(ctx) bind(C, name='drt_mark_cost_reduced')
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx
     end function drt_mark_cost_reduced
     integer(c_int) function drt_comm_unique_id(id) bind(C, name='drt_comm_unique_id')
       import :: c_int, c_char
       character(kind=c_char), intent(out) :: id(128)
     end function drt_comm_unique_id
     integer(c_int) function drt_comm_init(ctx, nranks, rank, id) bind(C, name='drt_comm_init')
       import :: c_ptr, c_int, c_int32_t, c_char
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: nranks, rank
       character(kind=c_char), intent(in) :: id(128)
     end function drt_comm_init
     integer(c_int) function drt_comm_destroy(ctx) bind(C, name='drt_comm_destroy')
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx
     end function drt_comm_destroy
     integer(c_int) function drt_comm_selftest(ctx) bind(C, name='drt_comm_selftest')
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx
     end function drt_comm_selftest
  end interface
end module dartray_c

#ifndef DRT_REAL_MODULES
!> Stand-in for the module state of smooth_grid_routines.f90 / rt_routines.f90: same names, kinds,
!> ranks and lower bounds (0-based as allocated by read_main_grid, io_routines.f90:329-332).
!> Built with -DDRT_REAL_MODULES (make linktest, build container only) the facade uses the REFERENCE's modules instead:
!> that build proves that every name, kind and rank the facade relies on exists there as declared here.
module dartray_state
  use iso_fortran_env
  implicit none
  type var_arr_2d                                    ! smooth_grid_routines.f90:110-112
     real(kind=real64), allocatable :: a(:,:)
  end type var_arr_2d
  integer(kind=int32) :: tot_ncell, max_lvl, tot_p_src, tot_ndir, tot_ndir_scaspe
  integer :: base(2), lnum
  integer(kind=int32), allocatable :: cchild(:), lvl(:), src_cell(:)
  integer(kind=int64), allocatable, target :: cindex(:)
  real(kind=real64), allocatable :: ccoord(:,:), csize(:), dens_arr(:,:), dens_stars_arr(:,:)
  real(kind=real64), allocatable :: u_fest_arr(:,:), u_final_arr(:,:), lumcell(:,:)
  real(kind=real64), allocatable :: lum_p_src_arr(:,:), ccoord_p_src(:,:), dir_i_out(:,:)
  real(kind=real64), allocatable :: ksca_arr_norm(:), gsca_arr(:), lum_lost(:)
  real(kind=real64), allocatable :: i_obs_arr(:,:,:), sed_arr(:,:), sed_arr_dir(:,:)
  type(var_arr_2d), allocatable :: scaspe_arr(:), scaspe_tot_arr(:)
  integer, allocatable :: npix_arr(:), npix_hp_arr(:), kp_sca_arr(:)
  integer(kind=int32) :: kp_sca_max, bm_par, bm_par_sca, bm_par_max      ! rt_routines.f90:9-18
  real(kind=real64) :: rad_lim, en_lim, conv_en_lim, accuracy, cs, dist_obs, modelsize
  logical :: x_wall_on, y_wall_on, z_wall_on, cnflag, done, limit_scattering_iterations
  real(kind=real64) :: x_wall_coord(2), y_wall_coord(2), z_wall_coord(2)
  integer :: rt_type, iterations, max_sca_iterations, np_mpi, id_mpi
  logical :: main_prc
  logical :: file_restore = .FALSE., file_restore_part1 = .FALSE.        ! io_routines.f90:137-140, set by check_files
  integer :: rt_algorithm_ID = 0                                           ! rt_routines.f90:48
  integer, parameter :: rta_main = 0, rta_i_obs = 1                        ! rt_routines.f90:236-238
  logical :: only_direct_rt = .FALSE., print_output_part2 = .FALSE.       ! rt_routines.f90:518, io_routines.f90:128
  logical :: print_psel_av = .FALSE.                                       ! rt_routines.f90:270
  real(kind=real64), allocatable :: psel_av_arr(:,:,:)                     ! smooth_grid_routines.f90:88
  ! rt_type ids, rt_routines.f90:195-217
  integer, parameter :: rtt_start = 0, rtt_precalc_cell = 1, rtt_precalc_src = 2, rtt_output_part1 = 3, &
       rtt_prep_part2 = 4, rtt_dir_cell = 5, rtt_i_obs_dir_cell = 6, rtt_dir_src = 7, rtt_i_obs_dir_src = 8, &
       rtt_output_part2 = 9, rtt_scatt = 10, rtt_i_obs = 11
end module dartray_state
#endif

!> The facade: dartray_hub.f90's driver procedures on top of the GPU library.
module dartray_hub_gpu
  use iso_c_binding
  use iso_fortran_env
  use dartray_c
#ifdef DRT_REAL_MODULES
  ! inside the reference: the module state itself (rt_routines re-exports smooth_grid_routines); only the data is
  ! imported - rt_routines' own rt_loop / rt_loop_iobs are the procedures this facade replaces
  use rt_routines, only: var_arr_2d, tot_ncell, max_lvl, tot_p_src, tot_ndir, tot_ndir_scaspe, base, lnum, cchild, lvl, src_cell, &
       cindex, ccoord, csize, dens_arr, dens_stars_arr, u_fest_arr, u_final_arr, lumcell, lum_p_src_arr, ccoord_p_src, &
       dir_i_out, ksca_arr_norm, gsca_arr, lum_lost, i_obs_arr, sed_arr, sed_arr_dir, scaspe_arr, scaspe_tot_arr, &
       npix_arr, npix_hp_arr, kp_sca_arr, kp_sca_max, bm_par, bm_par_sca, bm_par_max, rad_lim, en_lim, conv_en_lim, &
       accuracy, cs, dist_obs, modelsize, x_wall_on, y_wall_on, z_wall_on, cnflag, done, limit_scattering_iterations, &
       x_wall_coord, y_wall_coord, z_wall_coord, rt_type, iterations, max_sca_iterations, np_mpi, id_mpi, main_prc, &
       rtt_start, rtt_precalc_cell, rtt_precalc_src, rtt_output_part1, rtt_prep_part2, rtt_dir_cell, rtt_i_obs_dir_cell, &
       rtt_dir_src, rtt_i_obs_dir_src, rtt_output_part2, rtt_scatt, rtt_i_obs, rt_algorithm_ID, rta_main, rta_i_obs, only_direct_rt, &
       print_psel_av, psel_av_arr
#else
  use dartray_state
#endif
  implicit none
  type(c_ptr), save :: gpu_ctx = c_null_ptr
  integer :: gpu_device = 0
  abstract interface
     subroutine output_hook()
     end subroutine output_hook
  end interface
  !> Called where dartray_main calls make_output (dartray_hub.f90:27,43,52,59); the reference's HDF5
  !> writers (io_routines.f90:1743) plug in here, the module arrays are up to date at that point.
  procedure(output_hook), pointer :: make_output => null()
  !> Stage restart (dartray_hub.f90:397-417): check_files (io_routines.f90:893) decides what can be restored and sets
  !> file_restore / file_restore_part1; read_output (io_routines.f90:1900, case rtt_start) fills u_fest_arr, or u_final_arr,
  !> scaspe_arr, lum_lost and i_obs_arr, from the files of an earlier run. Left null, every run starts from the beginning.
  procedure(output_hook), pointer :: check_files => null(), read_output => null()
  !> dartray_i_obs: fills scaspe_tot_arr (HEALPix part) from the scaspe_tot files of the main run (read_output, case
  !> rtt_read_scaspe_tot, io_routines.f90:1988-1995); the arrays are allocated by the caller.
  procedure(output_hook), pointer :: read_scaspe_tot => null()
#ifdef DRT_REAL_MODULES
  ! io_routines.f90:137-140; that module needs the HDF5 Fortran modules, which amdflang cannot read in this image, so the
  ! link test keeps the two flags here
  logical :: file_restore = .FALSE., file_restore_part1 = .FALSE.
  logical :: print_output_part2 = .FALSE.   ! io_routines.f90:128, the same reason
#endif
contains

  !> Error convention of the reference: print + stop (stop_prc, smooth_grid_routines.f90:869-892).
  subroutine gpu_check(rc, where)
    integer(c_int), intent(in) :: rc
    character(len=*), intent(in) :: where
    character(kind=c_char), pointer :: msg(:)
    type(c_ptr) :: p
    integer :: i
    if (rc == 0) return
    p = drt_last_error(gpu_ctx)
    write(*, '(a,a,a,i0)', advance='no') 'STOP(', where, '): dartray_rt error ', rc
    if (c_associated(p)) then
       call c_f_pointer(p, msg, [512])
       write(*, '(a)', advance='no') ' : '
       do i = 1, 512
          if (msg(i) == c_null_char) exit
          write(*, '(a)', advance='no') msg(i)
       enddo
    endif
    write(*, *)
    stop 1
  end subroutine gpu_check

  !> Upload what grid_initialize (dartray_hub.f90:472-499) has prepared in module state.
  subroutine gpu_grid_initialize()
    type(drt_params) :: p
    integer(c_size_t) :: n
    call gpu_check(drt_create(gpu_ctx, int(gpu_device, c_int)), 'drt_create')
    call gpu_check(drt_set_grid(gpu_ctx, tot_ncell, max_lvl, int(base, c_int32_t), cchild, lvl, cindex, ccoord, csize), 'drt_set_grid')
    p%kp_sca_max = kp_sca_max; p%bm_par = bm_par; p%bm_par_sca = bm_par_sca; p%bm_par_max = bm_par_max
    p%rad_lim = rad_lim; p%accuracy = accuracy; p%conv_en_lim = conv_en_lim; p%cs = cs; p%dist_obs = dist_obs
    p%wall_on = 0
    if (x_wall_on) p%wall_on(1) = 1
    if (y_wall_on) p%wall_on(2) = 1
    if (z_wall_on) p%wall_on(3) = 1
    p%wall_coord(1:2) = x_wall_coord; p%wall_coord(3:4) = y_wall_coord; p%wall_coord(5:6) = z_wall_coord
    p%max_sca_iterations = -1
    if (limit_scattering_iterations) p%max_sca_iterations = max_sca_iterations
    p%np_mpi = np_mpi; p%id_mpi = id_mpi
    call gpu_check(drt_set_params(gpu_ctx, p), 'drt_set_params')
    call gpu_check(drt_set_wavelengths(gpu_ctx, int(lnum, c_int32_t), ksca_arr_norm, gsca_arr), 'drt_set_wavelengths')
    call gpu_check(drt_set_point_sources(gpu_ctx, tot_p_src, ccoord_p_src, lum_p_src_arr), 'drt_set_point_sources')
    call gpu_check(drt_set_observers(gpu_ctx, tot_ndir, dir_i_out), 'drt_set_observers')
    n = int(lnum, c_size_t)*int(tot_ncell, c_size_t)
    call gpu_check(drt_set_array(gpu_ctx, DRT_A_DENS, 0, dens_arr, n), 'dens_arr')
    call gpu_check(drt_set_array(gpu_ctx, DRT_A_DENS_STARS, 0, dens_stars_arr, n), 'dens_stars_arr')
    if (print_psel_av) call gpu_check(drt_set_psel_av(gpu_ctx, 1_c_int32_t), 'drt_set_psel_av')
  end subroutine gpu_grid_initialize

  !> np_mpi > 1: create the RCCL communicator rt_mpi reduces over. `id` is the 128-byte unique id that the main rank
  !> obtained with gpu_comm_unique_id and broadcast (MPI_Bcast of 128 MPI_BYTE after initialize_mpi,
  !> io_routines.f90:3610); np_mpi / id_mpi are the ones gpu_grid_initialize passed to drt_set_params.
  subroutine gpu_comm_unique_id(id)
    character(kind=c_char), intent(out) :: id(128)
    call gpu_check(drt_comm_unique_id(id), 'drt_comm_unique_id')
  end subroutine gpu_comm_unique_id

  subroutine gpu_comm_init(id)
    character(kind=c_char), intent(in) :: id(128)
    call gpu_check(drt_comm_init(gpu_ctx, int(np_mpi, c_int32_t), int(id_mpi, c_int32_t), id), 'drt_comm_init')
  end subroutine gpu_comm_init

  !> Memory for the ray queues in GiB and the number of persistent wavefronts (drt_configure; 0 = leave as it is). Not needed for a
  !> run: a context that is never configured sizes its queues by itself (DRT_ARENA_GB in the environment, else by the number of
  !> wavelengths and the device's memory, include/dartray_rt.h).
  subroutine gpu_configure(arena_gib, waves)
    real(c_double), intent(in) :: arena_gib
    integer, intent(in) :: waves
    call gpu_check(drt_configure(gpu_ctx, int(arena_gib * 1073741824.0_c_double, c_size_t), int(waves, c_int)), 'drt_configure')
  end subroutine gpu_configure

  !> The library's RCCL path on one rank (drt_comm_selftest): stops with the library's message when RCCL cannot be used here.
  subroutine gpu_comm_selftest()
    call gpu_check(drt_comm_selftest(gpu_ctx), 'drt_comm_selftest')
  end subroutine gpu_comm_selftest

  !> rt_prepare (dartray_hub.f90:386-469) for the cases that touch the path.
  subroutine rt_prepare()
    integer :: i
    integer(c_int32_t) :: a, b, c
    select case (rt_type)
    case (rtt_start)
       call gpu_check(drt_rt_prepare(gpu_ctx, rtt_start), 'rt_prepare(start)')
       en_lim = drt_en_lim(gpu_ctx)
       if (.not. allocated(src_cell)) allocate(src_cell(0:tot_ncell-1))
       call gpu_check(drt_get_src_cell(gpu_ctx, src_cell, ccoord_p_src), 'src_cell')
       if (.not. allocated(npix_arr)) allocate(npix_arr(0:lnum-1), npix_hp_arr(0:lnum-1), kp_sca_arr(0:lnum-1))
       do i = 0, lnum-1
          call gpu_check(drt_get_npix(gpu_ctx, int(i, c_int32_t), a, b, c), 'npix')
          npix_arr(i) = a; npix_hp_arr(i) = b; kp_sca_arr(i) = c
       enddo
       iterations = 0
       if (associated(check_files)) call check_files()
       if (file_restore_part1) then
          ! u_fest of an earlier run instead of the two precalculation stages; the stage machine goes on at prep_part2
          if (.not. allocated(u_fest_arr)) allocate(u_fest_arr(0:lnum-1, 0:tot_ncell-1))
          call read_output()
          call gpu_check(drt_set_array(gpu_ctx, DRT_A_U_FEST, 0, u_fest_arr, int(lnum, c_size_t)*int(tot_ncell, c_size_t)), 'u_fest_arr')
          rt_type = rtt_output_part1
       endif
       if (file_restore) then
          ! the direct-light fields of an earlier run; the stage machine goes on at the scattering iterations
          call gpu_check(drt_rt_prepare(gpu_ctx, rtt_prep_part2), 'rt_prepare(prep_part2)')     ! create_scaspe, create_i_obs
          call gpu_alloc_part2
          call read_output()
          call gpu_restore_part2
          rt_type = rtt_output_part2
       endif
    case (rtt_prep_part2)
       call gpu_check(drt_rt_prepare(gpu_ctx, rtt_prep_part2), 'rt_prepare(prep_part2)')
    case (rtt_scatt)
       call gpu_check(drt_rt_prepare(gpu_ctx, rtt_scatt), 'rt_prepare(scatt)')
       iterations = drt_iterations(gpu_ctx)
       cnflag = drt_cnflag(gpu_ctx) /= 0
       if (iterations == 1) bm_par = bm_par_sca
    case (rtt_i_obs)
       call gpu_check(drt_rt_prepare(gpu_ctx, rtt_i_obs), 'rt_prepare(i_obs)')
    case (rtt_i_obs_dir_cell, rtt_output_part2)
       ! nothing to do for the stellar-emission algorithm
    case default
       print *, 'you should not get here: rt_prepare'
       stop 1
    end select
  end subroutine rt_prepare

  !> The module arrays read_output fills when file_restore is set (create_scaspe, create_i_obs; u_final_arr and lum_lost
  !> exist from grid_initialize in the reference).
  subroutine gpu_alloc_part2()
    integer :: i
    if (.not. allocated(u_final_arr)) allocate(u_final_arr(0:lnum-1, 0:tot_ncell-1))
    if (.not. allocated(lum_lost)) allocate(lum_lost(0:lnum-1))
    if (.not. allocated(scaspe_arr)) then
       allocate(scaspe_arr(0:lnum-1))
       do i = 0, lnum-1
          allocate(scaspe_arr(i)%a(0:npix_arr(i)-1, 0:tot_ncell-1))
       enddo
    endif
    if (tot_ndir > 0 .and. .not. allocated(i_obs_arr)) allocate(i_obs_arr(0:lnum-1, 0:tot_ncell+tot_p_src-1, 0:tot_ndir-1))
  end subroutine gpu_alloc_part2

  !> Upload of what read_output restored (the reverse of gpu_sync_to_host(rtt_output_part2)).
  subroutine gpu_restore_part2()
    integer :: i
    integer(c_size_t) :: n
    n = int(lnum, c_size_t)*int(tot_ncell, c_size_t)
    call gpu_check(drt_set_array(gpu_ctx, DRT_A_U_FINAL, 0, u_final_arr, n), 'u_final_arr')
    call gpu_check(drt_set_array(gpu_ctx, DRT_A_LUM_LOST, 0, lum_lost, int(lnum, c_size_t)), 'lum_lost')
    do i = 0, lnum-1
       call gpu_check(drt_set_array(gpu_ctx, DRT_A_SCASPE, int(i, c_int32_t), scaspe_arr(i)%a, &
            int(npix_arr(i), c_size_t)*int(tot_ncell, c_size_t)), 'scaspe_arr')
    enddo
    if (tot_ndir > 0) call gpu_check(drt_set_array(gpu_ctx, DRT_A_I_OBS, 0, i_obs_arr, &
         int(lnum, c_size_t)*int(tot_ncell+tot_p_src, c_size_t)*int(tot_ndir, c_size_t)), 'i_obs_arr')
  end subroutine gpu_restore_part2

  !> rt_loop (rt_routines.f90:985): the source loop, ray fan, traversal and deposition, on the GPU.
  subroutine rt_loop()
    call gpu_check(drt_rt_loop(gpu_ctx, int(rt_type, c_int32_t)), 'rt_loop')
  end subroutine rt_loop

  !> rt_loop_iobs (rt_routines.f90:1470)
  subroutine rt_loop_iobs()
    call gpu_check(drt_rt_loop_iobs(gpu_ctx, int(rt_type, c_int32_t)), 'rt_loop_iobs')
  end subroutine rt_loop_iobs

  !> rt_mpi (io_routines.f90:3667)
  subroutine rt_mpi()
    call gpu_check(drt_rt_mpi(gpu_ctx, int(rt_type, c_int32_t)), 'rt_mpi')
  end subroutine rt_mpi

  !> psel_av_arr(0, iterations, :) of the stage just reduced (rt_routines.f90:1456-1465; create_psel_av_arr / expand_psel_av_arr
  !> :3219-3290 keep one row per scattering iteration): called after rt_mpi of the direct light and of every scattering iteration.
  subroutine gpu_collect_psel_av()
    real(kind=real64), allocatable :: tmp(:,:,:)
    integer :: m, nit
    if (.not. print_psel_av) return
    m = tot_ncell + tot_p_src
    if (.not. allocated(psel_av_arr)) then
       allocate(psel_av_arr(0:0, 0:iterations, 0:m-1))
       psel_av_arr = 0
    else if (size(psel_av_arr, 2) < iterations + 1) then
       nit = size(psel_av_arr, 2)
       allocate(tmp(0:0, 0:nit-1, 0:m-1))
       tmp = psel_av_arr
       deallocate(psel_av_arr)
       allocate(psel_av_arr(0:0, 0:iterations, 0:m-1))
       psel_av_arr = 0
       psel_av_arr(:, 0:nit-1, :) = tmp
       deallocate(tmp)
    endif
    call gpu_check(drt_get_array(gpu_ctx, DRT_A_PSEL_AV, 0, psel_av_arr(0, iterations, :), int(m, c_size_t)), 'psel_av_arr')
  end subroutine gpu_collect_psel_av

  !> Bring the fields a stage produced back into module state (what make_output and calc_sed read).
  subroutine gpu_sync_to_host(stage)
    integer, intent(in) :: stage
    integer :: i
    integer(c_size_t) :: n
    n = int(lnum, c_size_t)*int(tot_ncell, c_size_t)
    select case (stage)
    case (rtt_i_obs_dir_cell)      ! dartray_i_obs: the observer arrays alone
       if (tot_ndir > 0) then
          if (.not. allocated(i_obs_arr)) allocate(i_obs_arr(0:lnum-1, 0:tot_ncell+tot_p_src-1, 0:tot_ndir-1))
          call gpu_check(drt_get_array(gpu_ctx, DRT_A_I_OBS, 0, i_obs_arr, &
               int(lnum, c_size_t)*int(tot_ncell+tot_p_src, c_size_t)*int(tot_ndir, c_size_t)), 'i_obs_arr')
       endif
    case (rtt_output_part1)
       if (.not. allocated(u_fest_arr)) allocate(u_fest_arr(0:lnum-1, 0:tot_ncell-1))
       call gpu_check(drt_get_array(gpu_ctx, DRT_A_U_FEST, 0, u_fest_arr, n), 'u_fest_arr')
    case (rtt_output_part2, rtt_i_obs)
       if (.not. allocated(u_final_arr)) allocate(u_final_arr(0:lnum-1, 0:tot_ncell-1))
       call gpu_check(drt_get_array(gpu_ctx, DRT_A_U_FINAL, 0, u_final_arr, n), 'u_final_arr')
       if (.not. allocated(lum_lost)) allocate(lum_lost(0:lnum-1))
       call gpu_check(drt_get_array(gpu_ctx, DRT_A_LUM_LOST, 0, lum_lost, int(lnum, c_size_t)), 'lum_lost')
       if (stage == rtt_output_part2) then
          if (.not. allocated(scaspe_arr)) then
             allocate(scaspe_arr(0:lnum-1))
             do i = 0, lnum-1
                allocate(scaspe_arr(i)%a(0:npix_arr(i)-1, 0:tot_ncell-1))
             enddo
          endif
          do i = 0, lnum-1
             call gpu_check(drt_get_array(gpu_ctx, DRT_A_SCASPE, int(i, c_int32_t), scaspe_arr(i)%a, &
                  int(npix_arr(i), c_size_t)*int(tot_ncell, c_size_t)), 'scaspe_arr')
          enddo
       else
          if (.not. allocated(scaspe_tot_arr)) then
             allocate(scaspe_tot_arr(0:lnum-1))
             do i = 0, lnum-1
                allocate(scaspe_tot_arr(i)%a(0:npix_arr(i)-1, 0:tot_ncell-1))
             enddo
          endif
          do i = 0, lnum-1
             call gpu_check(drt_get_array(gpu_ctx, DRT_A_SCASPE_TOT, int(i, c_int32_t), scaspe_tot_arr(i)%a, &
                  int(npix_arr(i), c_size_t)*int(tot_ncell, c_size_t)), 'scaspe_tot_arr')
          enddo
       endif
       if (tot_ndir > 0) then
          if (.not. allocated(i_obs_arr)) allocate(i_obs_arr(0:lnum-1, 0:tot_ncell+tot_p_src-1, 0:tot_ndir-1))
          call gpu_check(drt_get_array(gpu_ctx, DRT_A_I_OBS, 0, i_obs_arr, &
               int(lnum, c_size_t)*int(tot_ncell+tot_p_src, c_size_t)*int(tot_ndir, c_size_t)), 'i_obs_arr')
       endif
    end select
  end subroutine gpu_sync_to_host

  !> calc_sed (sed_routines.f90:90-146) for the units 'erg/s/Hz/pc^2/sr'
  subroutine calc_sed()
    if (tot_ndir == 0) return
    if (.not. allocated(sed_arr)) then
       allocate(sed_arr(0:lnum-1, 0:tot_ndir-1), sed_arr_dir(0:lnum-1, 0:tot_ndir-1))
       sed_arr = 0; sed_arr_dir = 0
    endif
    select case (rt_type)
    case (rtt_output_part2)
       call gpu_check(drt_calc_sed(gpu_ctx, sed_arr_dir), 'calc_sed')
    case (rtt_i_obs)
       call gpu_check(drt_calc_sed(gpu_ctx, sed_arr), 'calc_sed')
    end select
  end subroutine calc_sed

  !> select_rt_type (dartray_hub.f90:314-382)
  subroutine select_rt_type()
    integer, parameter :: ntype = 12
    integer :: i, rt_type_arr(ntype)
    done = .FALSE.
    rt_type_arr = [rtt_start, rtt_precalc_cell, rtt_precalc_src, rtt_output_part1, rtt_prep_part2, rtt_dir_cell, &
         rtt_dir_src, rtt_i_obs_dir_cell, rtt_i_obs_dir_src, rtt_output_part2, rtt_scatt, rtt_i_obs]
    do i = 1, ntype
       if (rt_type == rt_type_arr(i)) exit
    enddo
    if (i < ntype) then
       rt_type = rt_type_arr(i+1)
    else
       done = .TRUE.
    endif
  end subroutine select_rt_type

  !> dartray_main (dartray_hub.f90:12-67): the stage machine; make_output / make_maps are the
  !> caller's business (hooks: gpu_sync_to_host fills the arrays they print).
  subroutine dartray_main()
    rt_type = rtt_start
    do
       select case (rt_type)
       case (rtt_start)
          call rt_prepare
       case (rtt_precalc_cell)
          call rt_loop
       case (rtt_precalc_src)
          call rt_loop
          call rt_mpi
       case (rtt_output_part1)
          call gpu_sync_to_host(rtt_output_part1)
          if (associated(make_output)) call make_output()
       case (rtt_prep_part2)
          call rt_prepare
       case (rtt_dir_cell)
          call rt_loop
       case (rtt_dir_src)
          call rt_loop
          call rt_mpi
          call gpu_collect_psel_av
       case (rtt_i_obs_dir_cell)
          call rt_prepare
          call rt_loop_iobs
       case (rtt_i_obs_dir_src)
          call rt_loop_iobs
       case (rtt_output_part2)
          call rt_prepare
          call calc_sed
          call gpu_sync_to_host(rtt_output_part2)
          if (associated(make_output)) call make_output()
       case (rtt_scatt)
          do
             call rt_prepare
             if (cnflag) exit
             call rt_loop
             call rt_mpi
             call gpu_collect_psel_av
          enddo
       case (rtt_i_obs)
          call rt_prepare
          call rt_loop_iobs
          call calc_sed
          call gpu_sync_to_host(rtt_i_obs)
          if (associated(make_output)) call make_output()
       end select
       call select_rt_type
       if (done) exit
    enddo
  end subroutine dartray_main

  !> dartray_i_obs (dartray_hub.f90:125-168), stellar emission: the observer passes for the directions of THIS run's
  !> file_dir_out on the fields of an earlier main run - direct light from the grid, scattered light from the HEALPix part
  !> of that run's scaspe_tot files (read_scaspe_tot hook). Output through make_output (reset_filenames' names).
  subroutine dartray_i_obs()
    integer :: i
    call gpu_check(drt_set_algorithm(gpu_ctx, int(rta_i_obs, c_int32_t)), 'drt_set_algorithm')
    rt_type = rtt_start                 ! prepare_p_src, calc_total_luminosity (:137-139)
    call rt_prepare
    rt_type = rtt_prep_part2            ! create_scaspe, create_i_obs (:138, :140)
    call rt_prepare
    rt_type = rtt_i_obs_dir_src
    call rt_loop_iobs
    rt_type = rtt_i_obs_dir_cell
    call rt_loop_iobs
    rt_type = rtt_output_part2
    if (print_output_part2) then
       call calc_sed
       call gpu_sync_to_host(rtt_i_obs_dir_cell)
       if (associated(make_output)) call make_output()
    endif
    if (only_direct_rt) return
    ! create_scaspe_tot + read_output(rtt_read_scaspe_tot) (:155-157)
    if (.not. associated(read_scaspe_tot)) then
       print *, 'STOP(dartray_i_obs): no reader for the scaspe_tot files (read_scaspe_tot)'
       stop 1
    endif
    if (.not. allocated(scaspe_tot_arr)) then
       allocate(scaspe_tot_arr(0:lnum-1))
       do i = 0, lnum-1
          allocate(scaspe_tot_arr(i)%a(0:npix_arr(i)-1, 0:tot_ncell-1))
          scaspe_tot_arr(i)%a = 0
       enddo
    endif
    call read_scaspe_tot()
    do i = 0, lnum-1
       call gpu_check(drt_set_array(gpu_ctx, DRT_A_SCASPE_TOT, int(i, c_int32_t), scaspe_tot_arr(i)%a, &
            int(npix_arr(i), c_size_t)*int(tot_ncell, c_size_t)), 'scaspe_tot_arr')
    enddo
    rt_type = rtt_i_obs
    call rt_prepare
    call rt_loop_iobs
    call calc_sed
    call gpu_sync_to_host(rtt_i_obs_dir_cell)
    if (associated(make_output)) call make_output()
  end subroutine dartray_i_obs

  subroutine gpu_finalize()
    integer(c_int) :: rc
    if (c_associated(gpu_ctx)) rc = drt_destroy(gpu_ctx)
    gpu_ctx = c_null_ptr
  end subroutine gpu_finalize
end module dartray_hub_gpu
