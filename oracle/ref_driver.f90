!> ORACLE INFRASTRUCTURE (test-only; never linked into or called from the product path).
!>
!> Drives the UNMODIFIED reference hot-path modules (smooth_grid_routines, healpix_routines,
!> ray_list, rt_routines), compiled from /root/reference where they lie by oracle/Makefile,
!> through the stage order of dartray_main (dartray_hub.f90:12-67). The reference's own driver
!> cannot be built here (io_routines.f90 needs a gfortran hdf5.mod, SURVEY.md section 8c), so the
!> few lines of glue that live in io_routines/dartray_hub are restated below, each with its
!> citation. Inputs come from a stream file written by dart-ray_amd/grids.py (Case.write);
!> outputs go to "<prefix>.out" and become the golden fixtures under tests/golden/.
!>
!> usage: ref_driver <case.bin> <out-prefix> [stages]
!>   stages: bitmask, 1=precalc 2=direct 4=iobs-direct 8=scattering 16=iobs-scatt (default 31),
!>           64 = print_psel_av: the average path of every source's rays per stage, written to "<prefix>.psel"
!>           32 = afterwards the i_obs algorithm (dartray_i_obs) for other observer directions, written to "<prefix>.iobs"
program ref_driver
  use smooth_grid_routines
  use healpix_routines
  use ray_list
  use rt_routines
  implicit none
  integer :: ierr, un, uo, i, j, k, nsrc, ndir, stages, magic, max_it
  integer(kind=int32) :: hdr(16)
  real(kind=real64) :: dbl(12)
  character(len=512) :: fcase, fout, arg
  real(kind=real64) :: t0, t1, t_pre, t_dir, t_sca, t_iobs
  real(kind=real64), allocatable :: sed_dir(:,:), sed_tot(:,:), tmp2(:,:)
  integer :: wall_i(3)

  call mpi_init(ierr)
  call mpi_comm_size(MPI_COMM_WORLD, np_mpi, ierr)
  call mpi_comm_rank(MPI_COMM_WORLD, id_mpi, ierr)
  main_prc = (id_mpi == 0)
  nproc = 1
  call get_environment_variable('OMP_NUM_THREADS', arg, status=ierr)
  if (ierr == 0) read(arg, *, iostat=ierr) nproc
  if (nproc < 1) nproc = 1

  call get_command_argument(1, fcase)
  call get_command_argument(2, fout)
  stages = 31
  if (command_argument_count() >= 3) then
     call get_command_argument(3, arg)
     read(arg, *) stages
  endif

  ! ---------------------------------------------------------------- read the case
  open(newunit=un, file=trim(fcase), access='stream', form='unformatted', status='old', action='read')
  read(un) hdr
  read(un) dbl
  magic = hdr(1)
  if (magic /= int(z'44525431')) stop 'ref_driver: bad case file'
  tot_ncell = hdr(2); max_lvl = hdr(3); base(1) = hdr(4); base(2) = hdr(5); lnum = hdr(6)
  nsrc = hdr(7); ndir = hdr(8); kp_sca_max = hdr(9)
  bm_par = hdr(10); bm_par_sca = hdr(11); bm_par_max = hdr(12)
  wall_i = hdr(13:15); max_it = hdr(16)
  modelsize = dbl(1); rad_lim = dbl(2); accuracy = dbl(3); conv_en_lim = dbl(4); dist_obs = dbl(5)
  cs = dbl(6)
  x_wall_on = wall_i(1) /= 0; y_wall_on = wall_i(2) /= 0; z_wall_on = wall_i(3) /= 0
  x_wall_coord = dbl(7:8); y_wall_coord = dbl(9:10); z_wall_coord = dbl(11:12)

  ! read_main_grid (io_routines.f90:329-332, 374-387): bounds and initial values
  allocate(cchild(0:tot_ncell-1), cindex(0:tot_ncell-1), lvl(0:tot_ncell-1), ccoord(3,0:tot_ncell-1), &
       csize(0:tot_ncell-1), src_cell(0:tot_ncell-1), lock_cell(0:tot_ncell-1))
  allocate(dens_arr(0:lnum-1,0:tot_ncell-1), dens_stars_arr(0:lnum-1,0:tot_ncell-1), &
       u_fest_arr(0:lnum-1,0:tot_ncell-1), u_final_arr(0:lnum-1,0:tot_ncell-1))
  read(un) cchild
  read(un) lvl
  read(un) cindex
  read(un) ccoord
  read(un) csize
  read(un) dens_arr
  read(un) dens_stars_arr
  lnum_stars = lnum; lnum_tot = lnum; lnum_dust = 0
  allocate(ksca_arr_norm(0:lnum-1), gsca_arr(0:lnum-1), lambda_arr(0:lnum-1))
  read(un) ksca_arr_norm
  read(un) gsca_arr
  tot_p_src = nsrc
  allocate(ccoord_p_src(3,0:max(nsrc,1)-1), cell_src(0:max(nsrc,1)-1), lum_p_src_arr(0:lnum-1,0:max(nsrc,1)-1))
  if (nsrc > 0) then
     read(un) ccoord_p_src
     read(un) lum_p_src_arr
  endif
  tot_ndir = ndir
  tot_ndir_scaspe = ndir      ! read_dir_out, io_routines.f90:559 (rt_algorithm 'main')
  tot_ndir_in = 0
  allocate(dir_i_out(0:max(ndir,1)-1,2))
  if (ndir > 0) read(un) dir_i_out
  close(un)
  do i = 0, lnum-1
     lambda_arr(i) = 1.0_real64 + i
  enddo

  u_fest_arr = 0; u_final_arr = 0; lock_cell = 0; src_cell = -1
  tot_spare_cells = 0; iterations = 0; iterations_dustem = 0

  ! input_initialize / check_input defaults that steer the path (io_routines.f90:2654-2742, 3187-3190)
  rt_algorithm = 'main'; rt_algorithm_ID = rta_main
  no_communications = .TRUE.; sequential_scattering = .TRUE.
  print_psel_av = (iand(stages,64) /= 0); print_scaspe_tot = .FALSE.; test_run = .FALSE.; only_direct_rt = .FALSE.
  use_lambda_grid = .FALSE.
  limit_scattering_iterations = (max_it >= 0); max_sca_iterations = max_it
  lnum_maps = 0
  allocate(ind_out_maps(0:0)); ind_out_maps = -1
  lnum_node = 0   ! the precalc stage relies on this being 0 before create_scaspe (SURVEY 8c caveat 4)
  dim_npix_unique = 0
  cnflag = .FALSE.

  ! grid_initialize (dartray_hub.f90:472-499), the parts that exist outside io_routines
  call set_base                                   ! smooth_grid_routines.f90:622
  ! make_csize_arr (io_routines.f90:669-696)
  allocate(csize_arr(0:max_lvl), carea_arr(0:max_lvl), cvol_arr(0:max_lvl))
  pabs_max = modelsize*10
  do i = 0, max_lvl
     if (i > 0) then
        csize_arr(i) = modelsize*(baseinv(1)*baseinv(2)**(i-1))
     else
        csize_arr(i) = modelsize
     endif
     carea_arr(i) = csize_arr(i)**2
     cvol_arr(i) = csize_arr(i)**3
  enddo
  ! set_chunk_size (io_routines.f90:3523-3537)
  chunk_size = tot_ncell/nproc**2/np_mpi
  if (chunk_size == 0) stop 'ref_driver: chunk_size = 0'
  call set_npix_arr                               ! rt_routines.f90:5235

  open(newunit=uo, file=trim(fout)//'.out', access='stream', form='unformatted', status='replace')
  write(uo) int(z'44524F31', kind=int32), tot_ncell, lnum, ndir, nsrc
  write(uo) npix_arr(0:lnum-1), npix_hp_arr(0:lnum-1), kp_sca_arr(0:lnum-1)

  ! ---------------------------------------------------------------- rtt_start (dartray_hub.f90:390-395)
  rt_type = rtt_start
  call set_en_lim
  call prepare_p_src
  call calc_total_luminosity
  write(uo) en_lim
  write(uo) src_cell
  if (nsrc > 0) write(uo) ccoord_p_src

  t_pre = 0; t_dir = 0; t_sca = 0; t_iobs = 0

  ! ---------------------------------------------------------------- precalculation
  if (iand(stages,1) /= 0) then
     t0 = mpi_wtime()
     rt_type = rtt_precalc_cell
     call rt_loop
     rt_type = rtt_precalc_src
     call rt_loop
     call drv_reduce_2d(u_fest_arr)                 ! reduce_u_fest_arr, io_routines.f90:3708-3729
     t_pre = mpi_wtime() - t0
  endif
  write(uo) u_fest_arr

  ! ---------------------------------------------------------------- prep_part2 (dartray_hub.f90:419-424)
  rt_type = rtt_prep_part2
  call create_scaspe
  call create_i_obs
  call pre_calc_ads_arr
  call create_psel_av_arr
  call create_mpi_type

  ! ---------------------------------------------------------------- direct light
  if (iand(stages,2) /= 0) then
     t0 = mpi_wtime()
     rt_type = rtt_dir_cell
     call rt_loop
     rt_type = rtt_dir_src
     call rt_loop
     call drv_rt_mpi
     t_dir = mpi_wtime() - t0
  endif
  write(uo) u_final_arr
  do i = 0, lnum-1
     write(uo) scaspe_arr(i)%a
  enddo
  write(uo) lum_lost

  allocate(sed_dir(0:lnum-1,0:max(ndir,1)-1), sed_tot(0:lnum-1,0:max(ndir,1)-1))
  sed_dir = 0; sed_tot = 0
  if (iand(stages,4) /= 0 .and. ndir > 0) then
     t0 = mpi_wtime()
     rt_type = rtt_i_obs_dir_cell
     call rt_loop_iobs
     rt_type = rtt_i_obs_dir_src
     call rt_loop_iobs
     t_iobs = t_iobs + mpi_wtime() - t0
     call drv_sed(sed_dir)
  endif
  if (ndir > 0) then
     write(uo) i_obs_arr
     write(uo) sed_dir
  endif

  ! ---------------------------------------------------------------- scattering iterations
  if (iand(stages,8) /= 0) then
     t0 = mpi_wtime()
     rt_type = rtt_scatt
     do
        ! rt_prepare, case rtt_scatt (dartray_hub.f90:432-457)
        u_fest_arr = u_final_arr
        lum_lost_prev = lum_lost
        iterations = iterations + 1
        if (iterations == 1) then
           call create_scaspe_tot
           call create_scaspe_prev
           bm_par = bm_par_sca
        endif
        do i = 0, lnum-1
           scaspe_prev_arr(i)%a = scaspe_arr(i)%a
        enddo
        call calc_total_luminosity_sca
        if (cnflag) exit
        call expand_psel_av_arr                       ! rt_prepare, dartray_hub.f90:450 (returns unless print_psel_av)
        call rt_loop
        call drv_rt_mpi
     enddo
     t_sca = mpi_wtime() - t0
     write(uo) iterations
     write(uo) u_final_arr
     do i = 0, lnum-1
        write(uo) scaspe_tot_arr(i)%a
     enddo
     write(uo) lum_lost

     if (iand(stages,16) /= 0 .and. ndir > 0) then
        t0 = mpi_wtime()
        rt_type = rtt_i_obs
        call calc_total_luminosity_sca
        call rt_loop_iobs
        t_iobs = t_iobs + mpi_wtime() - t0
        call drv_sed(sed_tot)
     endif
     if (ndir > 0) then
        write(uo) i_obs_arr
        write(uo) sed_tot
     endif

     ! ------------------------------------------------------------- dartray_i_obs (dartray_hub.f90:125-168)
     ! The i_obs algorithm for OTHER observer directions, on the scaspe_tot of this run (kept in memory where the reference
     ! reads it back from its file, :155-157). Its only effect on the path: rt_loop_iobs(rtt_i_obs) takes the scattered
     ! luminosity from the HEALPix pixel holding the direction (rt_routines.f90:1538-1548).
     if (iand(stages,32) /= 0 .and. ndir > 0) then
        rt_algorithm = 'i_obs'; rt_algorithm_ID = rta_i_obs
        do j = 0, tot_ndir-1
           dir_i_out(j,1) = 0.5_real64*dir_i_out(j,1) + 0.3_real64
           dir_i_out(j,2) = 0.5_real64*dir_i_out(j,2) + 0.4_real64
        enddo
        open(newunit=un, file=trim(fout)//'.iobs', access='stream', form='unformatted', status='replace')
        write(un) int(z'44524F32', kind=int32), tot_ncell, lnum, ndir, nsrc
        write(un) dir_i_out
        rt_type = rtt_start
        call calc_total_luminosity                      ! :139: lumcell back to the stellar emission
        call create_i_obs                               ! :140: zeroes i_obs_arr
        rt_type = rtt_i_obs_dir_src                     ! :141-144
        call rt_loop_iobs
        rt_type = rtt_i_obs_dir_cell
        call rt_loop_iobs
        call drv_sed(sed_dir)
        write(un) i_obs_arr
        write(un) sed_dir
        rt_type = rtt_i_obs                             ! :160-163
        call calc_total_luminosity_sca
        call rt_loop_iobs
        call drv_sed(sed_tot)
        write(un) i_obs_arr
        write(un) sed_tot
        close(un)
     endif
  endif
  write(uo) t_pre, t_dir, t_sca, t_iobs
  close(uo)
  if (print_psel_av) then
     ! psel_av_arr(0, 0:iterations, 0:tot_ncell+tot_p_src-1) (rt_routines.f90:1456-1465, 3219-3290)
     open(newunit=un, file=trim(fout)//'.psel', access='stream', form='unformatted', status='replace')
     write(un) int(z'44524F33', kind=int32), tot_ncell, nsrc, int(size(psel_av_arr,2), kind=int32)
     write(un) psel_av_arr
     close(un)
  endif
  if (main_prc) print '(a,4f10.3)', 'ref_driver timings pre/dir/sca/iobs [s]: ', t_pre, t_dir, t_sca, t_iobs

  call mpi_finalize(ierr)

contains

  !> MPI_Reduce(SUM)+Bcast of an (L,N) array == Allreduce (io_routines.f90:3708-3729)
  subroutine drv_reduce_2d(arr)
    real(kind=real64) :: arr(0:,0:)
    real(kind=real64), allocatable :: in_arr(:,:)
    integer :: ierr
    if (np_mpi == 1) return
    allocate(in_arr(0:size(arr,1)-1,0:size(arr,2)-1))
    call mpi_allreduce(arr, in_arr, size(arr), MPI_DOUBLE_PRECISION, MPI_SUM, MPI_COMM_WORLD, ierr)
    arr = in_arr
    deallocate(in_arr)
  end subroutine drv_reduce_2d

  !> rt_mpi for rtt_dir_src / rtt_scatt (io_routines.f90:3667-3704): reduce_u_final_arr (:3732-3761),
  !> reduce_scaspe_arr (:3806-3834), reduce_lum_lost (:3837-3855).
  subroutine drv_rt_mpi
    real(kind=real64), allocatable :: in_arr(:,:), in1(:)
    integer :: ierr, i
    allocate(in_arr(0:lnum-1,0:tot_ncell-1))
    in_arr = 0
    if (rt_type == rtt_scatt) then
       u_final_arr = u_final_arr - u_fest_arr
       where (u_final_arr < 0)
          u_final_arr = 0
       end where
    endif
    call mpi_allreduce(u_final_arr, in_arr, lnum*tot_ncell, MPI_DOUBLE_PRECISION, MPI_SUM, MPI_COMM_WORLD, ierr)
    if (rt_type /= rtt_scatt) then
       u_final_arr = in_arr
    else
       u_final_arr = in_arr + u_fest_arr
    endif
    deallocate(in_arr)
    if (np_mpi > 1) then
       do i = 0, lnum-1
          allocate(in_arr(0:npix_arr(i)-1,0:tot_ncell-1))
          in_arr = 0
          call mpi_allreduce(scaspe_arr(i)%a, in_arr, npix_arr(i)*tot_ncell, MPI_DOUBLE_PRECISION, MPI_SUM, MPI_COMM_WORLD, ierr)
          scaspe_arr(i)%a = in_arr
          deallocate(in_arr)
       enddo
    endif
    allocate(in1(0:lnum-1))
    in1 = 0
    if (rt_type == rtt_scatt) lum_lost = lum_lost - (np_mpi-1._real64)/np_mpi*lum_lost_prev
    call mpi_allreduce(lum_lost, in1, lnum, MPI_DOUBLE_PRECISION, MPI_SUM, MPI_COMM_WORLD, ierr)
    lum_lost = in1
    deallocate(in1)
  end subroutine drv_rt_mpi

  !> calc_sed + sum_i_obs for units 'erg/s/Hz/pc^2/sr' (sed_routines.f90:90-146, 211-246)
  subroutine drv_sed(sed)
    real(kind=real64) :: sed(0:,0:)
    real(kind=real64), allocatable :: io(:,:)
    integer :: il, j
    allocate(io(0:tot_ncell+tot_p_src-1,0:tot_ndir-1))
    do il = 0, lnum-1
       io = i_obs_arr(il,:,:)
       io = io*10.0_real64**(-7)
       io = io*10.0_real64**(26)/parsec**2.
       do j = 0, tot_ndir-1
          sed(il,j) = sum(io(0:tot_ncell-1,j)*(csize(:)/dist_obs)**2)
          if (tot_p_src > 0) then
             sed(il,j) = sed(il,j) + sum(io(tot_ncell:tot_ncell+tot_p_src-1,j)*(csize(cell_src)/dist_obs)**2)
          endif
       enddo
    enddo
    deallocate(io)
  end subroutine drv_sed

end program ref_driver
