!> Main program in the shape of dartray_trustI.f90:9-36 on the reference's own input files:
!>   dartray_gpu_main <input.in> [device]
!> reads the three input namelists, the wavelength list, the HDF5 main grid and lambda grids, observer
!> directions, point sources, the effective-opacity table and the source spectra, uploads the problem
!> (gpu_grid_initialize), runs dartray_main on the GPU and writes the HDF5 output files of make_output.
program dartray_gpu_main
  use iso_fortran_env
  use dartray_state
  use dartray_hub_gpu
  use dartray_io_gpu
  implicit none
  character(len=512) :: input_filename, arg

  call get_command_argument(1, input_filename)
  if (command_argument_count() >= 2) then
     call get_command_argument(2, arg)
     read(arg, *) gpu_device
  endif
  call input_initialize
  call read_input_file(input_filename)
  call read_lambda_list
  call check_input
  call grid_initialize
  if (command_argument_count() >= 3) then     ! dartray_gpu_main <input.in> <device> test_run
     call get_command_argument(3, arg)
     if (trim(arg) == 'test_run') test_run = .TRUE.
  endif
  if (test_run) then
     ! test_run of the reference (rt_routines.f90:492-493, io_routines.f90:3201-3203): everything but the RT loops
     print '(a,i0,a,i0,a,i0,a,i0)', 'dartray_gpu_main: test_run, tot_ncell ', tot_ncell, ' lnum ', lnum, ' point sources ', tot_p_src, ' directions ', tot_ndir
     print '(a,3es24.16)', 'dartray_gpu_main: dens_arr(first, ref, last wavelength) max ', maxval(dens_arr(0,:)), maxval(dens_ref), maxval(dens_arr(lnum-1,:))
     if (tot_p_src > 0) print '(a,2es24.16)', 'dartray_gpu_main: lum_p_src_arr(first, last wavelength) ', lum_p_src_arr(0,0), lum_p_src_arr(lnum-1,0)
     stop
  endif
  call gpu_grid_initialize
  if (rt_algorithm_ID == rta_i_obs) then
     ! dartray_i_obs.f90 of the reference: observer passes for this input's directions on the scaspe_tot files of a main run
     make_output => make_output_i_obs_hdf5
     read_scaspe_tot => read_scaspe_tot_hdf5
     call dartray_i_obs
     call gpu_finalize
     print '(a)', 'dartray_gpu_main: done (i_obs algorithm)'
     stop
  endif
  make_output => make_output_hdf5
  check_files => check_files_hdf5
  read_output => read_output_hdf5
  call dartray_main
  call gpu_finalize
  print '(a,i0,a)', 'dartray_gpu_main: done, ', iterations, ' scattering iterations'
end program dartray_gpu_main
