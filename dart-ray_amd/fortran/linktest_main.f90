!> Link test (make linktest): a driver in the shape of dartray_trustI.f90:9-36 whose module state is the REFERENCE's
!> (use rt_routines) and whose RT entry points are the GPU facade's. Compiled and linked, never run by the tests.
program linktest_main
  use rt_routines, only: tot_ncell, lnum, iterations
  use dartray_hub_gpu, only: gpu_grid_initialize, dartray_main, gpu_finalize
  implicit none
  if (command_argument_count() > 99) then   ! never true: the calls only have to link
     call gpu_grid_initialize
     call dartray_main
     call gpu_finalize
  endif
  print *, 'linktest: facade linked against the reference modules', tot_ncell, lnum, iterations
end program linktest_main
