!> dartray_io.f90 - the caller's side of the GPU path in the reference's own file formats
!> (SURVEY.md section 8f ranks 1 and 2): input namelists, wavelength list, observer directions, point
!> sources, effective opacities, HDF5 main / lambda grids in, HDF5 fields out. Names, namelists and file
!> layouts are those of io_routines.f90 so that an input file written for the reference is read as it is;
!> HDF5 goes through the C ABI of include/dartray_io.h (the image's hdf5.mod is gfortran's).
!>
!> Restated here (reference lines in each routine): input_initialize :2654, read_input_file :3991,
!> read_lambda_list :2258, read_dir_out :520, read_p_src :623, read_main_grid :297, read_lambda_grid :397,
!> set_walls, read_av_opacities (sed_routines.f90:1822), set_filenames :720, make_output :1743.
!> Not restated: dust-model preparation, pos_obs (internal observers), maps, restart files, dust emission.

module dartray_io_c
  use iso_c_binding
  implicit none
  interface
     integer(c_int) function drt_io_main_grid_size(path, tot_ncell) bind(C, name='drt_io_main_grid_size')
       import :: c_int, c_char, c_int32_t
       character(kind=c_char), intent(in) :: path(*)
       integer(c_int32_t), intent(out) :: tot_ncell
     end function drt_io_main_grid_size
     integer(c_int) function drt_io_read_main_grid(path, n, ncell, ccoord, cchild, cindex, lvl, csize, dens_ref, dens_stars_ref, base) &
          bind(C, name='drt_io_read_main_grid')
       import :: c_int, c_char, c_int32_t, c_int64_t, c_double
       character(kind=c_char), intent(in) :: path(*)
       integer(c_int32_t), value :: n
       integer(c_int32_t) :: ncell(*), cchild(*), lvl(*), base(2)
       integer(c_int64_t) :: cindex(*)
       real(c_double) :: ccoord(*), csize(*), dens_ref(*), dens_stars_ref(*)
     end function drt_io_read_main_grid
     integer(c_int) function drt_io_read_lambda_grid(path, n, dens, dens_stars) bind(C, name='drt_io_read_lambda_grid')
       import :: c_int, c_char, c_int32_t, c_double
       character(kind=c_char), intent(in) :: path(*)
       integer(c_int32_t), value :: n
       real(c_double) :: dens(*), dens_stars(*)
     end function drt_io_read_lambda_grid
     integer(c_int) function drt_io_write_array(path, dsetname, rank, fdims, data) bind(C, name='drt_io_write_array')
       import :: c_int, c_char, c_int64_t, c_double
       character(kind=c_char), intent(in) :: path(*), dsetname(*)
       integer(c_int), value :: rank
       integer(c_int64_t), intent(in) :: fdims(*)
       real(c_double), intent(in) :: data(*)
     end function drt_io_write_array
     integer(c_int) function drt_io_read_array(path, dsetname, rank, fdims, data) bind(C, name='drt_io_read_array')
       import :: c_int, c_char, c_int64_t, c_double
       character(kind=c_char), intent(in) :: path(*), dsetname(*)
       integer(c_int), value :: rank
       integer(c_int64_t), intent(in) :: fdims(*)
       real(c_double), intent(out) :: data(*)
     end function drt_io_read_array
     integer(c_int) function drt_io_array_dims(path, dsetname, rank, fdims) bind(C, name='drt_io_array_dims')
       import :: c_int, c_char, c_int64_t
       character(kind=c_char), intent(in) :: path(*), dsetname(*)
       integer(c_int), value :: rank
       integer(c_int64_t), intent(out) :: fdims(*)
     end function drt_io_array_dims
     integer(c_int) function drt_io_read_int(path, dsetname, v) bind(C, name='drt_io_read_int')
       import :: c_int, c_char, c_int32_t
       character(kind=c_char), intent(in) :: path(*), dsetname(*)
       integer(c_int32_t), intent(out) :: v
     end function drt_io_read_int
     integer(c_int) function drt_io_add_int(path, dsetname, v) bind(C, name='drt_io_add_int')
       import :: c_int, c_char, c_int32_t
       character(kind=c_char), intent(in) :: path(*), dsetname(*)
       integer(c_int32_t), value :: v
     end function drt_io_add_int
     integer(c_int) function drt_io_filename(out, cap, label, lambda, what, chext) bind(C, name='drt_io_filename')
       import :: c_int, c_char, c_size_t, c_double
       character(kind=c_char) :: out(*)
       integer(c_size_t), value :: cap
       character(kind=c_char), intent(in) :: label(*), what(*), chext(*)
       real(c_double), value :: lambda
     end function drt_io_filename
     integer(c_int) function drt_io_lambda_grid_filename(out, cap, label, lambda) bind(C, name='drt_io_lambda_grid_filename')
       import :: c_int, c_char, c_size_t, c_double
       character(kind=c_char) :: out(*)
       integer(c_size_t), value :: cap
       character(kind=c_char), intent(in) :: label(*)
       real(c_double), value :: lambda
     end function drt_io_lambda_grid_filename
     function drt_io_last_error() bind(C, name='drt_io_last_error')
       import :: c_ptr
       type(c_ptr) :: drt_io_last_error
     end function drt_io_last_error
  end interface
end module dartray_io_c

module dartray_io_gpu
  use iso_c_binding
  use iso_fortran_env
  use dartray_io_c
  use dartray_state
  use dartray_hub_gpu, only: calc_sed
  implicit none
  integer, parameter :: lcar = 200, max_n_input = 1000
  ! ---- the three input namelists (io_routines.f90:176-178); every name is declared so that any input file of
  ! the reference parses, the ones this path does not use are read and ignored
  character(len=lcar) :: label_model_lambda_grid, label_model_out, label_model_out_i_obs, file_dir_out, file_pos_obs, &
       file_p_src, file_lambda_list, dir_runs, dir_grid, grid_file, rt_algorithm, units_luminosity, units_csize, &
       units_lambda, dust_model, dust_opacity_tables, file_gra_fa, file_sil_fa, file_pah_neu_fa, file_pah_ion_fa, &
       file_av_opacities, dust_heating_type, file_q_gra, file_q_sil, file_q_pah_neu, file_q_pah_ion, &
       file_calorimetry_Gra, file_calorimetry_Sil, file_nbody_sph, file_stellar_library, stellar_library, &
       param_to_project, file_param_src
  real(kind=real64) :: lambda_ref, max_lambda_stars, min_lambda_dust, tau_cell_max, npixel_maps, map_size_factor, z_sun
  integer :: ind_i_obs(0:max_n_input-1), ind_out_maps(0:max_n_input-1), n_dust_size_qabs(0:3), n_dust_wave_qabs, &
       n_dust_temp_cal(0:3), kp_maps, n_int_rf_bins
  logical :: print_scaspe_tot, print_output_part1, print_scaspe_part2, restore_file_mpi, &   ! print_output_part2, only_direct_rt: dartray_state
       use_lambda_grid, use_dir_out, use_pos_obs, use_p_src, sequential_scattering, print_sed, &   ! print_psel_av: dartray_state
       input_av_opacities, no_communications, no_dust_rt, test_run, print_maps, print_maps_in, &
       use_stellar_library
  namelist /dartray_input_strings/ label_model_lambda_grid, label_model_out, label_model_out_i_obs, file_dir_out, &
       file_pos_obs, file_p_src, file_lambda_list, dir_runs, dir_grid, grid_file, rt_algorithm, units_luminosity, &
       units_csize, units_lambda, dust_model, dust_opacity_tables, file_gra_fa, file_sil_fa, file_pah_neu_fa, &
       file_pah_ion_fa, file_av_opacities, dust_heating_type, file_q_gra, file_q_sil, file_q_pah_neu, file_q_pah_ion, &
       file_calorimetry_Gra, file_calorimetry_Sil, file_nbody_sph, file_stellar_library, stellar_library, &
       param_to_project, file_param_src
  namelist /dartray_input_var/ kp_sca_max, rad_lim, accuracy, conv_en_lim, bm_par, bm_par_sca, bm_par_max, lambda_ref, &
       max_lambda_stars, min_lambda_dust, dist_obs, ind_i_obs, ind_out_maps, n_dust_size_qabs, n_dust_wave_qabs, &
       tau_cell_max, n_dust_temp_cal, npixel_maps, map_size_factor, kp_maps, x_wall_coord, y_wall_coord, z_wall_coord, &
       z_sun, max_sca_iterations, n_int_rf_bins
  namelist /dartray_input_logical/ print_scaspe_tot, print_output_part1, print_output_part2, print_scaspe_part2, &
       restore_file_mpi, use_lambda_grid, use_dir_out, use_pos_obs, use_p_src, print_psel_av, sequential_scattering, &
       print_sed, input_av_opacities, no_communications, no_dust_rt, only_direct_rt, test_run, print_maps, print_maps_in, &
       x_wall_on, y_wall_on, z_wall_on, use_stellar_library, limit_scattering_iterations
  ! ---- state the readers fill
  integer :: lnum_tot
  real(kind=real64), allocatable :: lambda_arr(:), dens_ref(:), dens_stars_ref(:)
  real(kind=real64), allocatable :: kext_arr(:)   ! extinction coefficient per wavelength (read_av_opacities)
  real(kind=real64) :: kext_ref                   ! ... and at lambda_ref
  integer(kind=int32), allocatable :: ncell(:)
  real(kind=real64), parameter :: cspeed = 2.99792458E8, parsec = 3.08567758E16   ! smooth_grid_routines.f90 (single-precision literals)
contains

  subroutine stop_prc(msg)
    character(len=*), intent(in) :: msg
    print *, trim(msg)
    stop 1
  end subroutine stop_prc

  subroutine io_check(rc, where)
    integer(c_int), intent(in) :: rc
    character(len=*), intent(in) :: where
    character(kind=c_char), pointer :: msg(:)
    type(c_ptr) :: p
    integer :: i
    if (rc == 0) return
    p = drt_io_last_error()
    write(*, '(a,a,a,i0,a)', advance='no') 'STOP(', where, '): dartray_io error ', rc, ' : '
    if (c_associated(p)) then
       call c_f_pointer(p, msg, [512])
       do i = 1, 512
          if (msg(i) == c_null_char) exit
          write(*, '(a)', advance='no') msg(i)
       enddo
    endif
    write(*, *)
    stop 1
  end subroutine io_check

  function cstr(s) result(c)
    character(len=*), intent(in) :: s
    character(kind=c_char) :: c(len_trim(adjustl(s))+1)
    character(len=len_trim(adjustl(s))) :: t
    integer :: i
    t = trim(adjustl(s))
    do i = 1, len(t)
       c(i) = t(i:i)
    enddo
    c(len(t)+1) = c_null_char
  end function cstr

  !> input_initialize (io_routines.f90:2654-2742): defaults of everything the namelists may set
  subroutine input_initialize()
    label_model_lambda_grid = 'not_provided'; label_model_out = 'not_provided'; label_model_out_i_obs = 'not_provided'
    file_dir_out = 'not_provided'; file_pos_obs = 'not_provided'; file_p_src = 'not_provided'
    file_lambda_list = 'not_provided'; dir_runs = 'not_provided'; dir_grid = 'not_provided'; grid_file = 'not_provided'
    rt_algorithm = 'not_provided'; units_luminosity = 'not_provided'; units_csize = 'not_provided'; units_lambda = 'not_provided'
    dust_model = 'not_provided'; dust_opacity_tables = 'not_provided'; file_av_opacities = 'not_provided'
    dust_heating_type = 'not_provided'; file_param_src = 'not_provided'; param_to_project = 'not_provided'
    file_gra_fa = 'not_provided'; file_sil_fa = 'not_provided'; file_pah_neu_fa = 'not_provided'; file_pah_ion_fa = 'not_provided'
    file_q_gra = 'not_provided'; file_q_sil = 'not_provided'; file_q_pah_neu = 'not_provided'; file_q_pah_ion = 'not_provided'
    file_calorimetry_Gra = 'not_provided'; file_calorimetry_Sil = 'not_provided'; file_nbody_sph = 'not_provided'
    file_stellar_library = 'not_provided'; stellar_library = 'not_provided'
    kp_sca_max = 0; rad_lim = 0; accuracy = 0; conv_en_lim = 0; bm_par = 0; bm_par_sca = 0; bm_par_max = 0
    lambda_ref = 0; max_lambda_stars = 0; min_lambda_dust = 0; dist_obs = 0; ind_i_obs = -1; ind_out_maps = -1
    n_dust_size_qabs = 0; n_dust_wave_qabs = 0; tau_cell_max = 0; n_dust_temp_cal = 0; npixel_maps = 0; map_size_factor = 0
    kp_maps = 0; x_wall_coord = 0; y_wall_coord = 0; z_wall_coord = 0; z_sun = 0; max_sca_iterations = 0; n_int_rf_bins = 0
    print_scaspe_tot = .FALSE.; print_output_part1 = .FALSE.; print_output_part2 = .FALSE.; print_scaspe_part2 = .FALSE.
    restore_file_mpi = .FALSE.; use_lambda_grid = .FALSE.; use_dir_out = .FALSE.; use_pos_obs = .FALSE.; use_p_src = .FALSE.
    print_psel_av = .FALSE.; sequential_scattering = .TRUE.; print_sed = .FALSE.; input_av_opacities = .FALSE.
    no_communications = .TRUE.; no_dust_rt = .TRUE.; only_direct_rt = .FALSE.; test_run = .FALSE.; print_maps = .FALSE.
    print_maps_in = .FALSE.; x_wall_on = .FALSE.; y_wall_on = .FALSE.; z_wall_on = .FALSE.; use_stellar_library = .FALSE.
    limit_scattering_iterations = .FALSE.
  end subroutine input_initialize

  !> read_input_file (io_routines.f90:3991-4019)
  subroutine read_input_file(input_filename)
    character(len=*), intent(in) :: input_filename
    integer :: ioun
    if (trim(input_filename) == '') call stop_prc('provide input file in command line...EXIT')
    open(newunit=ioun, file=trim(input_filename), status='old', action='read')
    read(ioun, nml=dartray_input_strings)
    read(ioun, nml=dartray_input_var)
    read(ioun, nml=dartray_input_logical)
    close(ioun)
  end subroutine read_input_file

  subroutine count_lines(unit, n)
    integer, intent(in) :: unit
    integer, intent(out) :: n
    integer :: v
    n = 0
    do
       read(unit, *, iostat=v)
       if (v /= 0) exit
       n = n + 1
    enddo
    rewind(unit)
  end subroutine count_lines

  !> read_lambda_list (io_routines.f90:2258-2350): the list has no comment lines; the stellar-emission RT uses
  !> the wavelengths up to max_lambda_stars (lnum_stars, :2339-2349)
  subroutine read_lambda_list()
    integer :: unit, i
    if (dir_grid == 'not_provided') call stop_prc('ERROR: Input dir_grid missing. Did you assign dir_grid in the input files ?')
    if (file_lambda_list == 'not_provided') call stop_prc('ERROR: Input file_lambda_list missing')
    open(newunit=unit, file=trim(adjustl(dir_grid))//trim(adjustl(file_lambda_list)), status='old')
    call count_lines(unit, lnum_tot)
    allocate(lambda_arr(0:lnum_tot-1))
    do i = 0, lnum_tot-1
       read(unit, *) lambda_arr(i)
    enddo
    close(unit)
    lnum = count(lambda_arr <= max_lambda_stars)
    if (lnum <= 0) call stop_prc('ERROR(read_lambda_list): no wavelength below max_lambda_stars')
  end subroutine read_lambda_list

  !> read_main_grid (io_routines.f90:297-394)
  subroutine read_main_grid()
    character(len=2*lcar) :: path
    integer(c_int32_t) :: n, b(2)
    path = trim(adjustl(dir_grid))//trim(adjustl(grid_file))
    call io_check(drt_io_main_grid_size(cstr(path), n), 'read_main_grid')
    tot_ncell = n
    allocate(ncell(0:tot_ncell-1), cchild(0:tot_ncell-1), cindex(0:tot_ncell-1), lvl(0:tot_ncell-1), ccoord(3,0:tot_ncell-1), &
         csize(0:tot_ncell-1), dens_ref(0:tot_ncell-1), dens_stars_ref(0:tot_ncell-1))
    allocate(dens_arr(0:lnum-1,0:tot_ncell-1), dens_stars_arr(0:lnum-1,0:tot_ncell-1))
    call io_check(drt_io_read_main_grid(cstr(path), n, ncell, ccoord, cchild, cindex, lvl, csize, dens_ref, dens_stars_ref, b), &
         'read_main_grid')
    base = b
    dens_arr = 0; dens_stars_arr = 0
    max_lvl = maxval(lvl)
    iterations = 0
  end subroutine read_main_grid

  !> read_lambda_grid (io_routines.f90:397-490): file names from :2327
  subroutine read_lambda_grid()
    integer :: iw
    character(kind=c_char) :: fname(512)
    character(len=512) :: f
    real(kind=real64), allocatable :: d(:), s(:)
    logical :: found_ref_grid
    integer :: i
    allocate(d(0:tot_ncell-1), s(0:tot_ncell-1))
    found_ref_grid = .FALSE.
    do iw = 0, lnum-1
       call io_check(drt_io_lambda_grid_filename(fname, int(512, c_size_t), cstr(label_model_lambda_grid), lambda_arr(iw)), 'read_lambda_grid')
       f = ''
       do i = 1, 512
          if (fname(i) == c_null_char) exit
          f(i:i) = fname(i)
       enddo
       call io_check(drt_io_read_lambda_grid(cstr(trim(adjustl(dir_grid))//trim(f)), tot_ncell, d, s), 'read_lambda_grid')
       dens_arr(iw,:) = d
       dens_stars_arr(iw,:) = s
       if (count(s < 0) > 0) call stop_prc('STOP(read_lambda_grid): negative value in dens_stars!')
       if (abs(lambda_ref-lambda_arr(iw))/lambda_ref < 1E-4) then
          dens_ref = d
          found_ref_grid = .TRUE.
       endif
    enddo
    if (.not. found_ref_grid) call stop_prc('ERROR(read_lambda_grid): reference grid not found among the lambda grids!')
  end subroutine read_lambda_grid

  !> read_dir_out (io_routines.f90:520-575): first line is a comment, then theta phi per line
  subroutine read_dir_out()
    integer :: id, num, i, j
    tot_ndir = 0; tot_ndir_scaspe = 0
    if (.not. use_dir_out) then
       allocate(dir_i_out(0:0,2))
       return
    endif
    open(newunit=id, file=trim(adjustl(dir_grid))//trim(adjustl(file_dir_out)), status='old')
    call count_lines(id, num)
    tot_ndir = num - 1
    if (tot_ndir <= 0) call stop_prc('something wrong while reading '//trim(adjustl(file_dir_out)))
    allocate(dir_i_out(0:tot_ndir-1,2))
    read(id, *)
    do i = 0, tot_ndir-1
       read(id, *) (dir_i_out(i,j), j=1,2)
    enddo
    close(id)
    tot_ndir_scaspe = tot_ndir
  end subroutine read_dir_out

  !> read_p_src (io_routines.f90:623-666): first line is a comment, then x y z per line
  subroutine read_p_src()
    integer :: id, num, i, j
    tot_p_src = 0
    if (.not. use_p_src) then
       allocate(ccoord_p_src(3,0:0), lum_p_src_arr(0:lnum-1,0:0))
       return
    endif
    open(newunit=id, file=trim(adjustl(dir_grid))//trim(adjustl(file_p_src)), status='old')
    call count_lines(id, num)
    tot_p_src = num - 1
    if (tot_p_src <= 0) call stop_prc('something wrong while reading '//trim(adjustl(file_p_src)))
    allocate(ccoord_p_src(3,0:tot_p_src-1), lum_p_src_arr(0:lnum-1,0:tot_p_src-1))
    read(id, *)
    do i = 0, tot_p_src-1
       read(id, *) (ccoord_p_src(j,i), j=1,3)
    enddo
    close(id)
    lum_p_src_arr = 0
  end subroutine read_p_src

  !> set_walls (rt_routines.f90): wall coordinates are given in units of modelsize from the lower model edge
  subroutine set_walls()
    modelsize = csize(0)
    if (x_wall_on) x_wall_coord = x_wall_coord*modelsize - modelsize/2.
    if (y_wall_on) y_wall_coord = y_wall_coord*modelsize - modelsize/2.
    if (z_wall_on) z_wall_coord = z_wall_coord*modelsize - modelsize/2.
  end subroutine set_walls

  !> set_units (rt_routines.f90:5180-5200): light speed in the length unit of the grid
  subroutine set_units()
    select case (trim(adjustl(units_csize)))
    case ('pc')
       cs = cspeed/parsec
    case default
       call stop_prc('ERROR(set_units): units_csize not recognized: '//trim(units_csize))
    end select
  end subroutine set_units

  !> read_av_opacities (sed_routines.f90:1822-1998): 4 header lines, then lambda[um] Cabs Csca tau/N_H albedo g;
  !> linear interpolation to lambda_arr between the first tabulated wavelength >= lambda and its predecessor
  !> (:1870-1898). Sets ksca_arr_norm = ksca/kext and gsca_arr for the lnum stellar wavelengths.
  subroutine read_av_opacities()
    real(kind=real64), allocatable :: tlambda(:), tkabs(:), tksca(:), tkext(:), tgsca(:)
    real(kind=real64) :: ttau, a2, m, c, kext, ksca
    integer :: u, nlines, opnum, i, k
    integer, parameter :: nhead = 4
    if (.not. input_av_opacities) call stop_prc('ERROR: this driver needs input_av_opacities = .TRUE. (no grain-size integration here)')
    open(newunit=u, file=trim(adjustl(file_av_opacities)), status='old')
    call count_lines(u, nlines)
    opnum = nlines - nhead
    allocate(tlambda(0:opnum-1), tkabs(0:opnum-1), tksca(0:opnum-1), tkext(0:opnum-1), tgsca(0:opnum-1))
    do i = 1, nhead
       read(u, *)
    enddo
    do i = 0, opnum-1
       read(u, *) tlambda(i), tkabs(i), tksca(i), ttau, a2, tgsca(i)
       if (i > 0) then
          if (tlambda(i) < tlambda(i-1)) call stop_prc('STOP: lambda not in ascending order in '//trim(file_av_opacities))
       endif
    enddo
    close(u)
    tkext = tkabs + tksca
    allocate(ksca_arr_norm(0:lnum-1), gsca_arr(0:lnum-1), kext_arr(0:lnum-1))
    k = 0
    do i = 0, opnum-1
       if (k > lnum-1) exit
       if (tlambda(i) < lambda_arr(k)) cycle
       if (i == 0) call stop_prc('STOP(read_av_opacities): wavelength below the tabulated range')
       m = (tkext(i) - tkext(i-1))/(tlambda(i)-tlambda(i-1)); c = tkext(i) - m*tlambda(i); kext = m*lambda_arr(k)+c
       m = (tksca(i) - tksca(i-1))/(tlambda(i)-tlambda(i-1)); c = tksca(i) - m*tlambda(i); ksca = m*lambda_arr(k)+c
       m = (tgsca(i) - tgsca(i-1))/(tlambda(i)-tlambda(i-1)); c = tgsca(i) - m*tlambda(i); gsca_arr(k) = m*lambda_arr(k)+c
       ksca_arr_norm(k) = ksca/kext
       kext_arr(k) = kext*1E-4     ! [cm^2] -> [m^2] (:1912); only ratios to kext_ref are used
       k = k + 1
    enddo
    if (k < lnum) call stop_prc('STOP(read_av_opacities): wavelength above the tabulated range')
    ! interpolate to lambda_ref (:1904-1908, value_locate :1662-1722)
    if (lambda_ref <= tlambda(0)) then
       i = 1
    else if (lambda_ref >= tlambda(opnum-1)) then
       i = opnum-1
    else
       do i = 1, opnum-1
          if (lambda_ref > tlambda(i-1) .and. lambda_ref <= tlambda(i)) exit
       enddo
    endif
    m = (tkext(i) - tkext(i-1))/(tlambda(i)-tlambda(i-1)); c = tkext(i) - m*tlambda(i)
    kext_ref = (m*lambda_ref+c)*1E-4
  end subroutine read_av_opacities

  !> scale_dens_arr (rt_routines.f90:5018-5071): without lambda grids dens_arr(lambda) is the main grid's dens (the
  !> extinction coefficient at lambda_ref) times kext(lambda)/kext_ref; with them, the two must agree within 5E-3 in
  !> every dusty leaf cell ("Did you use the same dust model when creating the grids ?").
  subroutine scale_dens_arr()
    integer :: i, j
    real(kind=real64), allocatable :: temp_arr(:)
    dens_ref = dens_ref/(kext_ref)
    if (.not. use_lambda_grid) then
       do i = 0, lnum-1
          dens_arr(i,:) = dens_ref(:)*kext_arr(i)
       enddo
    else
       allocate(temp_arr(0:tot_ncell-1))
       do i = 0, lnum-1
          temp_arr = dens_ref(:)*kext_arr(i)
          do j = 0, tot_ncell-1
             if (cchild(j) /= -1 .or. dens_ref(j) == 0) cycle
             if (abs(temp_arr(j)-dens_arr(i,j))/temp_arr(j) > 5E-3) then
                print *, 'STOP(scale_dens_arr): the input lambda grid dust density and that calculated by rescaling the ', &
                     'reference grid values do not match! Did you use the same dust model when creating the grids ?'
                print *, 'lambda = ', lambda_arr(i)
                print *, 'cell ID = ', j
                print *, 'lambda grid value = ', dens_arr(i,j)
                print *, 'scaled value = ', temp_arr(j)
                stop 1
             endif
          enddo
       enddo
       deallocate(temp_arr)
    endif
    dens_ref = dens_ref*kext_ref
  end subroutine scale_dens_arr

  !> check_input (io_routines.f90:2745-3303), the checks that concern this path: the algorithm, the files grid_initialize
  !> is going to open, units, and the allowed ranges of the RT parameters (:3015-3125). All problems are listed
  !> before stopping (error_stop).
  subroutine check_input()
    logical :: error, ex
    error = .FALSE.
    select case (trim(adjustl(rt_algorithm)))
    case ('main')
       rt_algorithm_ID = rta_main
    case ('i_obs')                       ! io_routines.f90:2778-2779
       rt_algorithm_ID = rta_i_obs
       if (label_model_out_i_obs == 'not_provided') label_model_out_i_obs = label_model_out   ! :2898-2902
       if (.not. use_dir_out) then
          print *, 'ERROR: rt_algorithm = ''i_obs'' needs observer directions (use_dir_out, file_dir_out)'; error = .TRUE.
       endif
    case ('not_provided')
       print *, 'ERROR: Input rt_algorithm missing'; error = .TRUE.
    case default
       print *, 'ERROR: rt_algorithm = ', trim(rt_algorithm), ' is outside the GPU path (stellar-emission RT: rt_algorithm = ''main'' or ''i_obs'')'
       error = .TRUE.
    end select
    if (dir_runs == 'not_provided') then
       print *, 'ERROR: Input dir_runs missing'; error = .TRUE.
    endif
    if (grid_file == 'not_provided') then
       print *, 'ERROR: Input grid_file missing'; error = .TRUE.
    else
       inquire(file=trim(adjustl(dir_grid))//trim(adjustl(grid_file)), exist=ex)
       if (.not. ex) then
          print *, 'ERROR: file not found: ', trim(adjustl(dir_grid))//trim(adjustl(grid_file)); error = .TRUE.
       endif
    endif
    if (use_dir_out) call need_file(dir_grid, file_dir_out, 'file_dir_out', error)
    if (use_p_src) call need_file(dir_grid, file_p_src, 'file_p_src', error)
    if (use_lambda_grid .and. label_model_lambda_grid == 'not_provided') then
       print *, 'ERROR: use_lambda_grid = .TRUE. needs label_model_lambda_grid'; error = .TRUE.
    endif
    if (label_model_out == 'not_provided') then
       print *, 'ERROR: Input label_model_out missing'; error = .TRUE.
    endif
    if (.not. input_av_opacities) then
       print *, 'ERROR: this driver needs input_av_opacities = .TRUE. and file_av_opacities (effective opacities; the grain-size ', &
            'integration of prepare_dust_model is outside the GPU path)'
       error = .TRUE.
    else
       call need_file('', file_av_opacities, 'file_av_opacities', error)
    endif
    if (lambda_ref <= 0) then
       print *, 'ERROR: Invalid lambda_ref value'; print *, 'lambda_ref = ', lambda_ref; print *, 'Allowed range = (0,Inf]'
       error = .TRUE.
    endif
    select case (trim(adjustl(units_luminosity)))
    case ('erg/s/Hz', 'W/Hz')
    case ('not_provided')
       print *, 'ERROR: input units_luminosity missing!'; error = .TRUE.
    case default
       print *, 'ERROR: units_luminosity not recognized! Possible choices: erg/s/Hz, W/Hz.'; error = .TRUE.
    end select
    select case (trim(adjustl(units_csize)))
    case ('pc')
    case ('not_provided')
       print *, 'ERROR: input units_csize missing!'; error = .TRUE.
    case default
       print *, 'ERROR: units_csize not recognized! Possible choices are: pc.'; error = .TRUE.
    end select
    if (error) call stop_prc('STOP(check_input): see the errors above')
    ! ---- parameters (:3015-3125)
    if (kp_sca_max < 0 .or. kp_sca_max > 4) then
       print *, 'ERROR: Invalid kp_sca_max value'; print *, 'kp_sca_max = ', kp_sca_max; print *, 'Allowed range = [0,4]'
       error = .TRUE.
    endif
    if (rad_lim < 0 .or. rad_lim > 2) then
       print *, 'ERROR: Invalid rad_lim value'; print *, 'rad_lim =', rad_lim; print *, 'Allowed range = [0, 2]'
       error = .TRUE.
    endif
    if (accuracy < 0 .or. accuracy > 1) then
       print *, 'ERROR: Invalid accuracy value'; print *, 'accuracy = ', accuracy; print *, 'Allowed range = [0,1]'
       error = .TRUE.
    endif
    if (conv_en_lim < 0. .or. conv_en_lim > 1) then
       print *, 'ERROR: Invalid conv_en_lim value'; print *, 'conv_en_lim = ', conv_en_lim; print *, 'Allowed range = [0,1]'
       error = .TRUE.
    endif
    if (bm_par < 0 .or. bm_par > 1000) then
       print *, 'ERROR: Invalid bm_par value'; print *, 'bm_par = ', bm_par; print *, 'Allowed range = [0,1000]'
       error = .TRUE.
    endif
    if (bm_par_sca < 0 .or. bm_par_sca > 1000) then
       print *, 'ERROR: Invalid bm_par_sca value'; print *, 'bm_par_sca = ', bm_par_sca; print *, 'Allowed range = [0,1000]'
       error = .TRUE.
    endif
    if (bm_par_max < 10*maxval([bm_par, bm_par_sca])) then
       print *, 'ERROR: Invalid bm_par_max value'; print *, 'bm_par_max = ', bm_par_max
       print *, 'Allowed range = [10*max([bm_par,bm_par_sca]), inf]'
       error = .TRUE.
    endif
    if (limit_scattering_iterations) then
       if (max_sca_iterations < 0 .or. max_sca_iterations > 10**6) then
          print *, 'ERROR: Invalid max_sca_iterations value'; print *, 'max_sca_iterations = ', max_sca_iterations
          print *, 'Allowed range = [0,1E6]'
          error = .TRUE.
       endif
       print *, 'WARNING: limit_scattering_iteration = .TRUE. The maximum number of scattering iterations will be :', max_sca_iterations
    endif
    if (print_sed) then
       if (dist_obs <= 0.1) then
          print *, 'ERROR: Invalid dist_obs'; print *, 'dist_obs = ', dist_obs; print *, 'Allowed range = (0.1, Inf]'
          error = .TRUE.
       endif
    endif
    ! walls (:3243-3275)
    if (x_wall_on) call check_wall(x_wall_coord, 'X', error)
    if (y_wall_on) call check_wall(y_wall_coord, 'y', error)
    if (z_wall_on) call check_wall(z_wall_coord, 'z', error)
    if (error) call stop_prc('STOP(check_input): see the errors above')
    ! ---- what the reference would go on to do and this program does not
    if (no_communications .and. .not. sequential_scattering) sequential_scattering = .TRUE.   ! (:3187-3190)
    if (.not. no_communications) print *, 'WARNING: the communication mode of the reference is not built; ranks hold the whole grid (no_communications)'
    if (.not. no_dust_rt) print *, 'WARNING: no_dust_rt = .FALSE.: the dust-emission RT is outside the GPU path, the run ends after the stellar-emission RT'
    if (print_maps) print *, 'WARNING: print_maps = .TRUE.: surface-brightness maps are outside the GPU path and will not be written'
    if (use_pos_obs) print *, 'WARNING: use_pos_obs = .TRUE.: internal observers are outside the GPU path and are ignored'
  contains
    subroutine need_file(dir, name, what, err)
      character(len=*), intent(in) :: dir, name, what
      logical, intent(inout) :: err
      logical :: there
      if (name == 'not_provided') then
         print *, 'ERROR: Input ', what, ' missing'; err = .TRUE.
         return
      endif
      inquire(file=trim(adjustl(dir))//trim(adjustl(name)), exist=there)
      if (.not. there) then
         print *, 'ERROR: file not found: ', trim(adjustl(dir))//trim(adjustl(name)); err = .TRUE.
      endif
    end subroutine need_file
    subroutine check_wall(w, ax, err)
      real(kind=real64), intent(in) :: w(2)
      character(len=*), intent(in) :: ax
      logical, intent(inout) :: err
      if (w(1) > w(2) .or. (w(1) < 0) .or. (w(2) > 1)) then
         print *, 'ERROR: ', ax, '_wall_coord values not allowed!'; print *, ax, '_wall_coord = ', w
         err = .TRUE.
      endif
    end subroutine check_wall
  end subroutine check_input

  !> Point-source spectra: the reference leaves lum_p_src_arr to the model's user routine (set_trustI,
  !> user_routines_trustI.f90:253-280: log-log interpolation of a two-column table, :131-156). Here the table
  !> named by file_param_src has one header line and columns lambda, L_nu(source 0), L_nu(source 1), ...
  subroutine read_param_src()
    real(kind=real64), allocatable :: tl(:), tlum(:,:)
    real(kind=real64) :: lam, m_t, c_t
    integer :: u, nlines, n, i, j, il, iq(1)
    logical :: bb_there
    real(kind=real64) :: l2
    if (tot_p_src == 0) return
    if (file_param_src == 'not_provided') then
       ! the TRUST-I main program (dartray_trustI.f90:27, set_trustI, user_routines_trustI.f90:250-285) takes the spectrum of
       ! its one star from <dir_grid>BB_T10000_L100000.dat: lambda [um], L_nu [erg/s/Hz], L_lambda; read_planck_table_trustI
       ! (:95-156) skips six header lines and interpolates log-log
       inquire(file=trim(adjustl(dir_grid))//'BB_T10000_L100000.dat', exist=bb_there)
       if (.not. bb_there .or. tot_p_src /= 1) call stop_prc('ERROR: use_p_src needs file_param_src (source spectra table)')
       open(newunit=u, file=trim(adjustl(dir_grid))//'BB_T10000_L100000.dat', status='old')
       call count_lines(u, nlines)
       n = nlines - 7
       allocate(tl(0:n-1), tlum(0:0, 0:n-1))
       do i = 1, 6
          read(u, *)
       enddo
       do i = 0, n-1
          read(u, *) tl(i), tlum(0,i), l2
       enddo
       close(u)
    else
       open(newunit=u, file=trim(adjustl(dir_grid))//trim(adjustl(file_param_src)), status='old')
       call count_lines(u, nlines)
       n = nlines - 1
       allocate(tl(0:n-1), tlum(0:tot_p_src-1, 0:n-1))
       read(u, *)
       do i = 0, n-1
          read(u, *) tl(i), (tlum(j,i), j = 0, tot_p_src-1)
       enddo
       close(u)
    endif
    do il = 0, lnum-1
       lam = lambda_arr(il)
       if (lam < tl(0)) call stop_prc('lambda smaller than tabulated lambda in table')
       if (lam > tl(n-1)) call stop_prc('lambda greater than tabulated lambda in table')
       iq = minloc(abs(tl-lam)) - 1
       if (lam <= tl(iq(1))) iq = iq - 1
       j = max(iq(1), 0)
       do i = 0, tot_p_src-1
          m_t = (log10(tlum(i,j+1))-log10(tlum(i,j)))/(log10(tl(j+1))-log10(tl(j)))
          c_t = log10(tlum(i,j+1)) - m_t*log10(tl(j+1))
          lum_p_src_arr(il,i) = 10.**(m_t*log10(lam)+c_t)
       enddo
    enddo
  end subroutine read_param_src

  !> grid_initialize (dartray_hub.f90:472-499), the reading part
  subroutine grid_initialize()
    call read_main_grid
    if (use_lambda_grid) then
       call read_lambda_grid
    else if (any(dens_stars_ref /= 0)) then
       ! without lambda grids the reference's user routines give the stellar emissivity its spectrum (e.g. set_galaxy)
       call stop_prc('ERROR: the main grid holds stellar emissivity: its spectrum needs use_lambda_grid = .TRUE. (dens_stars per wavelength)')
    endif
    call read_dir_out
    call read_p_src
    call set_units
    call read_av_opacities      ! prepare_dust_model with input_av_opacities: kext_arr, kext_ref, albedo, g
    call scale_dens_arr
    call read_param_src
    call set_walls
    np_mpi = 1; id_mpi = 0; main_prc = .TRUE.
  end subroutine grid_initialize

  !> set_filenames (io_routines.f90:720-794); int_names: the names reset_filenames (:797-830) gives the observer arrays and SEDs
  !> of the i_obs algorithm ('grid_<label_model_out_i_obs>_..._int.h5')
  subroutine out_name(lambda, what, path, int_names)
    real(kind=real64), intent(in) :: lambda
    character(len=*), intent(in) :: what
    character(len=*), intent(out) :: path
    logical, intent(in), optional :: int_names
    character(kind=c_char) :: fname(512)
    integer :: i
    logical :: alt
    alt = .FALSE.
    if (present(int_names)) alt = int_names
    if (alt) then
       call io_check(drt_io_filename(fname, int(512, c_size_t), cstr(label_model_out_i_obs), lambda, cstr(what), cstr('_int')), 'reset_filenames')
    else
       call io_check(drt_io_filename(fname, int(512, c_size_t), cstr(label_model_out), lambda, cstr(what), cstr('')), 'set_filenames')
    endif
    path = trim(adjustl(dir_runs))//'/'
    do i = 1, 512
       if (fname(i) == c_null_char) exit
       path = trim(path)//fname(i)
    enddo
  end subroutine out_name

  logical function out_exists(lambda, what)
    real(kind=real64), intent(in) :: lambda
    character(len=*), intent(in) :: what
    character(len=3*lcar) :: path
    call out_name(lambda, what, path)
    inquire(file=trim(path), exist=out_exists)
  end function out_exists

  !> check_files (io_routines.f90:893-1102): a final output file that exists already stops the run; otherwise, with
  !> restore_file_mpi set, a complete set of part-2 files (u_final, scaspe, i_obs, lum_lost after the direct light) makes
  !> the run start at the scattering iterations (file_restore), and failing that a complete set of part-1 files (u_fest)
  !> makes it start at the direct light (file_restore_part1).
  subroutine check_files_hdf5()
    integer :: i
    logical :: all_there
    file_restore = .FALSE.; file_restore_part1 = .FALSE.
    do i = 0, lnum-1
       if (out_exists(lambda_arr(i), 'ufield')) call stop_prc('STOP(check_files): the u_final output file exists already')
       if (use_dir_out) then
          if (out_exists(lambda_arr(i), 'i_obs')) call stop_prc('STOP(check_files): the i_obs output file exists already')
       endif
       if (print_scaspe_tot) then
          if (out_exists(lambda_arr(i), 'scaspe_tot')) call stop_prc('STOP(check_files): the scaspe_tot output file exists already')
       endif
    enddo
    if (out_exists(-1.0_real64, 'lum_lost')) call stop_prc('STOP(check_files): the lum_lost output file exists already')
    if (print_psel_av) then
       if (out_exists(-1.0_real64, 'psel_av')) call stop_prc('STOP(check_files): the psel_av output file exists already')
    endif
    all_there = .TRUE.
    do i = 0, lnum-1
       if (.not. out_exists(lambda_arr(i), 'ufield_part2')) all_there = .FALSE.
       if (.not. out_exists(lambda_arr(i), 'scaspe_part2')) all_there = .FALSE.
       if (use_dir_out) then
          if (.not. out_exists(lambda_arr(i), 'i_obs_part2')) all_there = .FALSE.
       endif
    enddo
    if (.not. out_exists(-1.0_real64, 'lum_lost_part2')) all_there = .FALSE.
    if (print_psel_av) then
       if (.not. out_exists(-1.0_real64, 'psel_av_part2')) all_there = .FALSE.
    endif
    if (all_there .and. restore_file_mpi) then
       file_restore = .TRUE.
       print *, 'Intermediate files will be restored'
       return
    endif
    all_there = .TRUE.
    do i = 0, lnum-1
       if (.not. out_exists(lambda_arr(i), 'u_fest_part1')) all_there = .FALSE.
    enddo
    if (all_there .and. restore_file_mpi) then
       file_restore_part1 = .TRUE.
       print *, 'Output part 1 will be restored!'
    endif
  end subroutine check_files_hdf5

  !> read_output, case rtt_start (io_routines.f90:1907-1957) through read_big_array (:2003): the inverse of
  !> make_output_hdf5 for the part-1 or part-2 files; the arrays are allocated by the caller (rt_prepare).
  subroutine read_output_hdf5()
    character(len=3*lcar) :: path
    integer :: i
    integer(c_int64_t) :: d1(1), d2(2)
    real(kind=real64), allocatable :: tmp(:), tmp2(:,:)
    allocate(tmp(0:tot_ncell-1))
    d1 = tot_ncell
    if (file_restore) then
       print *, 'restoring output part2...'
       do i = 0, lnum-1
          call out_name(lambda_arr(i), 'ufield_part2', path)
          call io_check(drt_io_read_array(cstr(path), cstr('u_final'), 1, d1, tmp), 'read_ufield_arr')
          u_final_arr(i,:) = tmp
          d2 = [int(npix_arr(i), c_int64_t), int(tot_ncell, c_int64_t)]
          call out_name(lambda_arr(i), 'scaspe_part2', path)
          call io_check(drt_io_read_array(cstr(path), cstr('scaspe'), 2, d2, scaspe_arr(i)%a), 'read_scaspe_arr')
          if (tot_ndir > 0) then
             allocate(tmp2(0:tot_ncell+tot_p_src-1, 0:tot_ndir-1))
             d2 = [int(tot_ncell+tot_p_src, c_int64_t), int(tot_ndir, c_int64_t)]
             call out_name(lambda_arr(i), 'i_obs_part2', path)
             call io_check(drt_io_read_array(cstr(path), cstr('i_obs'), 2, d2, tmp2), 'read_i_obs_arr')
             i_obs_arr(i,:,:) = tmp2
             deallocate(tmp2)
          endif
       enddo
       d1 = lnum
       call out_name(-1.0_real64, 'lum_lost_part2', path)
       call io_check(drt_io_read_array(cstr(path), cstr('lum_lost'), 1, d1, lum_lost), 'read_lum_lost')
       if (print_psel_av) then      ! read_psel_av_arr (io_routines.f90:1364-1379)
          if (allocated(psel_av_arr)) deallocate(psel_av_arr)
          allocate(psel_av_arr(0:0, 0:0, 0:tot_ncell+tot_p_src-1))
          call out_name(-1.0_real64, 'psel_av_part2', path)
          call io_check(drt_io_read_array(cstr(path), cstr('psel_av_arr'), 3, &
               [1_c_int64_t, 1_c_int64_t, int(tot_ncell+tot_p_src, c_int64_t)], psel_av_arr), 'read_psel_av_arr')
       endif
    endif
    if (file_restore_part1) then
       print *, 'restoring output part 1...'
       do i = 0, lnum-1
          call out_name(lambda_arr(i), 'u_fest_part1', path)
          call io_check(drt_io_read_array(cstr(path), cstr('u_fest'), 1, d1, tmp), 'read_u_fest_arr')
          u_fest_arr(i,:) = tmp
       enddo
    endif
  end subroutine read_output_hdf5

  !> read_output, case rtt_read_scaspe_tot (io_routines.f90:1988-1995) + find_out_tot_ndir_scaspe (:2566-2651): the scaspe_tot
  !> files of the main run hold npix_hp + (that run's number of observer directions) rows per cell; the HEALPix rows go to
  !> scaspe_tot_arr, whose remaining rows (this run's directions) stay zero - the i_obs algorithm never reads them.
  subroutine read_scaspe_tot_hdf5()
    character(len=3*lcar) :: path
    integer :: i, nhp
    integer(c_int32_t) :: nside_read
    integer(c_int64_t) :: d2(2)
    real(kind=real64), allocatable :: tmp2(:,:)
    do i = 0, lnum-1
       call out_name(lambda_arr(i), 'scaspe_tot', path)
       call io_check(drt_io_array_dims(cstr(path), cstr('scaspe_tot'), 2, d2), 'find_out_tot_ndir_scaspe')
       call io_check(drt_io_read_int(cstr(path), cstr('nside_sca'), nside_read), 'find_out_tot_ndir_scaspe')
       nhp = 1
       if (nside_read /= 0) nhp = 12*nside_read**2
       if (nhp /= npix_hp_arr(i) .or. d2(2) /= tot_ncell .or. d2(1) < nhp) then
          print *, 'error(read_big_array): the dimensions of the scaspe_tot file do not fit this run. Same kp_sca_max and grid as in the main RT calculation ?'
          print *, 'file dims =', d2, ' nside_sca =', nside_read, ' expected npix_hp =', npix_hp_arr(i), ' cells =', tot_ncell
          stop 1
       endif
       allocate(tmp2(0:d2(1)-1, 0:tot_ncell-1))
       call io_check(drt_io_read_array(cstr(path), cstr('scaspe_tot'), 2, d2, tmp2), 'read_scaspe_tot_arr')
       scaspe_tot_arr(i)%a = 0
       scaspe_tot_arr(i)%a(0:nhp-1, :) = tmp2(0:nhp-1, :)
       deallocate(tmp2)
    enddo
  end subroutine read_scaspe_tot_hdf5

  !> make_output for the i_obs algorithm (io_routines.f90:1743-1897 with the names of reset_filenames): the observer arrays and
  !> SEDs of the direct light (rtt_output_part2, when print_output_part2) and of the total light (rtt_i_obs).
  subroutine make_output_i_obs_hdf5()
    character(len=3*lcar) :: path
    integer :: i
    integer(c_int64_t) :: d2(2)
    real(kind=real64), allocatable :: tmp2(:,:)
    logical :: part2
    if (tot_ndir == 0) return
    part2 = (rt_type == rtt_output_part2)
    allocate(tmp2(0:tot_ncell+tot_p_src-1, 0:tot_ndir-1))
    d2 = [int(tot_ncell+tot_p_src, c_int64_t), int(tot_ndir, c_int64_t)]
    do i = 0, lnum-1
       tmp2 = i_obs_arr(i,:,:)
       if (part2) then
          call out_name(lambda_arr(i), 'i_obs_part2', path, .TRUE.)
       else
          call out_name(lambda_arr(i), 'i_obs', path, .TRUE.)
       endif
       call io_check(drt_io_write_array(cstr(path), cstr('i_obs'), 2, d2, tmp2), 'print_i_obs_arr')
    enddo
    if (print_sed) then
       d2 = [int(lnum, c_int64_t), int(tot_ndir, c_int64_t)]
       if (part2) then
          call out_name(-1.0_real64, 'sed_dir', path, .TRUE.)
          call io_check(drt_io_write_array(cstr(path), cstr('sed_arr_dir'), 2, d2, sed_arr_dir), 'print_sed_arr_dir')
       else
          call out_name(-1.0_real64, 'sed', path, .TRUE.)
          call io_check(drt_io_write_array(cstr(path), cstr('sed_arr'), 2, d2, sed_arr), 'print_sed_arr')
       endif
    endif
  end subroutine make_output_i_obs_hdf5

  !> make_output (io_routines.f90:1743-1897) through print_big_array (:1112): one HDF5 file per array and wavelength
  subroutine make_output_hdf5()
    character(len=3*lcar) :: path
    integer :: i
    integer(c_int64_t) :: d1(1), d2(2)
    real(kind=real64), allocatable :: tmp(:), tmp2(:,:)
    select case (rt_type)
    case (rtt_output_part1)
       if (.not. print_output_part1) return
       allocate(tmp(0:tot_ncell-1))
       d1 = tot_ncell
       do i = 0, lnum-1
          tmp = u_fest_arr(i,:)
          call out_name(lambda_arr(i), 'u_fest_part1', path)
          call io_check(drt_io_write_array(cstr(path), cstr('u_fest'), 1, d1, tmp), 'print_u_fest_arr')
       enddo
    case (rtt_output_part2, rtt_i_obs)
       if (rt_type == rtt_output_part2 .and. .not. print_output_part2) return
       allocate(tmp(0:tot_ncell-1))
       d1 = tot_ncell
       do i = 0, lnum-1
          tmp = u_final_arr(i,:)
          if (rt_type == rtt_output_part2) then
             call out_name(lambda_arr(i), 'ufield_part2', path)
          else
             call out_name(lambda_arr(i), 'ufield', path)
          endif
          call io_check(drt_io_write_array(cstr(path), cstr('u_final'), 1, d1, tmp), 'print_ufield_arr')
          d2 = [int(npix_arr(i), c_int64_t), int(tot_ncell, c_int64_t)]
          if (rt_type == rtt_output_part2) then
             if (print_scaspe_part2) then
                call out_name(lambda_arr(i), 'scaspe_part2', path)
                call io_check(drt_io_write_array(cstr(path), cstr('scaspe'), 2, d2, scaspe_arr(i)%a), 'print_scaspe_arr')
             endif
          else if (print_scaspe_tot) then
             call out_name(lambda_arr(i), 'scaspe_tot', path)
             call io_check(drt_io_write_array(cstr(path), cstr('scaspe_tot'), 2, d2, scaspe_tot_arr(i)%a), 'print_scaspe_tot_arr')
             call io_check(drt_io_add_int(cstr(path), cstr('nside_sca'), int(2**kp_sca_arr(i), c_int32_t)), 'print_scaspe_tot_arr')
          endif
          if (tot_ndir > 0) then
             allocate(tmp2(0:tot_ncell+tot_p_src-1, 0:tot_ndir-1))
             tmp2 = i_obs_arr(i,:,:)
             d2 = [int(tot_ncell+tot_p_src, c_int64_t), int(tot_ndir, c_int64_t)]
             if (rt_type == rtt_output_part2) then
                call out_name(lambda_arr(i), 'i_obs_part2', path)
             else
                call out_name(lambda_arr(i), 'i_obs', path)
             endif
             call io_check(drt_io_write_array(cstr(path), cstr('i_obs'), 2, d2, tmp2), 'print_i_obs_arr')
             deallocate(tmp2)
          endif
       enddo
       if (print_psel_av .and. allocated(psel_av_arr)) then
          ! print_psel_av_arr (io_routines.f90:1337-1361): (1, 1, sources) after the direct light, (1, iterations, sources) at the end
          if (rt_type == rtt_output_part2) then
             call out_name(-1.0_real64, 'psel_av_part2', path)
             call io_check(drt_io_write_array(cstr(path), cstr('psel_av_arr'), 3, &
                  [1_c_int64_t, 1_c_int64_t, int(tot_ncell+tot_p_src, c_int64_t)], psel_av_arr(0:0, 0:0, :)), 'print_psel_av_arr')
          else
             call out_name(-1.0_real64, 'psel_av', path)
             call io_check(drt_io_write_array(cstr(path), cstr('psel_av_arr'), 3, &
                  [1_c_int64_t, int(min(iterations, size(psel_av_arr, 2)), c_int64_t), int(tot_ncell+tot_p_src, c_int64_t)], &
                  psel_av_arr(0:0, 0:min(iterations, size(psel_av_arr, 2))-1, :)), 'print_psel_av_arr')
          endif
       endif
       d1 = lnum
       if (rt_type == rtt_output_part2) then
          call out_name(-1.0_real64, 'lum_lost_part2', path)
       else
          call out_name(-1.0_real64, 'lum_lost', path)
       endif
       call io_check(drt_io_write_array(cstr(path), cstr('lum_lost'), 1, d1, lum_lost), 'print_lum_lost')
       if (print_sed .and. tot_ndir > 0) then
          d2 = [int(lnum, c_int64_t), int(tot_ndir, c_int64_t)]
          if (rt_type == rtt_output_part2) then
             call out_name(-1.0_real64, 'sed_dir', path)
             call io_check(drt_io_write_array(cstr(path), cstr('sed_arr_dir'), 2, d2, sed_arr_dir), 'print_sed_arr_dir')
          else
             call out_name(-1.0_real64, 'sed', path)
             call io_check(drt_io_write_array(cstr(path), cstr('sed_arr'), 2, d2, sed_arr), 'print_sed_arr')
          endif
       endif
    end select
  end subroutine make_output_hdf5
end module dartray_io_gpu
