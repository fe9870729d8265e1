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
  call grid_initialize
  call gpu_grid_initialize
  make_output => make_output_hdf5
  call dartray_main
  call gpu_finalize
  print '(a,i0,a)', 'dartray_gpu_main: done, ', iterations, ' scattering iterations'
end program dartray_gpu_main
