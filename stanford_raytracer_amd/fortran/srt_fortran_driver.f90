! srt_fortran_driver.f90 -- a Fortran main program that drives the HIP path through srt_bindc: the shape a
! maintainer's replacement for the ray loop of raytracer_driver.f95:1144-1232 takes.
!   srt_fortran_driver <newray.in> <rays.txt> <out.ray>       (Ngo model, fixed RK4, 100 steps)
program srt_fortran_driver
  use iso_c_binding
  use srt_bindc
  implicit none
  character(len=1000) :: cfg, raysfile, outfile
  type(c_ptr) :: model, ppos, pdir, pw
  type(srt_params) :: p
  real(c_double), pointer :: pos0(:,:), dir0(:,:), w0(:)
  real(c_double), allocatable :: rows(:)
  integer(c_int32_t), allocatable :: nrows(:), stopcond(:)
  real(c_double) :: qs(4), ms(4)
  integer(c_int64_t) :: nrays, steps
  integer(c_int) :: rc
  integer :: slots

  call get_command_argument(1, cfg)
  call get_command_argument(2, raysfile)
  call get_command_argument(3, outfile)
  rc = srt_init(0_c_int)
  if (rc /= 0) stop 'srt_init failed (an MI355X is required)'
  rc = srt_model_create_ngo(trim(cfg)//c_null_char, 2010001_c_int, 0_c_int, model)
  if (rc /= 0) stop 'srt_model_create_ngo failed'
  nrays = srt_read_rays_file(trim(raysfile)//c_null_char, ppos, pdir, pw)
  if (nrays < 0) stop 'cannot read the ray file'
  call c_f_pointer(ppos, pos0, (/ 3, int(nrays) /))
  call c_f_pointer(pdir, dir0, (/ 3, int(nrays) /))
  call c_f_pointer(pw, w0, (/ int(nrays) /))
  p%dt0 = 1.0e-3_c_double; p%dtmax = 0.1_c_double; p%tmax = 0.1_c_double; p%maxerr = 5.0e-4_c_double
  p%minalt = 6.4712e6_c_double; p%del = 1.0e-4_c_double
  p%maxsteps = 2000; p%root = 2; p%fixedstep = 1; p%outputper = 25; p%first_attempt_policy = 0; p%refill_threshold = 0
  p%ray_order = 0
  slots = srt_rows_per_ray(p)
  allocate(rows(SRT_ROW*slots*nrays), nrows(nrays), stopcond(nrays))
  rc = srt_trace_batch(model, p, nrays, pos0, dir0, w0, rows, nrows, stopcond, steps)
  if (rc /= 0) stop 'srt_trace_batch failed'
  rc = srt_model_species(model, qs, ms)
  rc = srt_write_ray_file(trim(outfile)//c_null_char, 0_c_int, 1_c_int64_t, nrays, p, srt_model_nspec(model), qs, ms, &
       w0, rows, nrows, stopcond)
  if (rc /= 0) stop 'srt_write_ray_file failed'
  print *, 'rays:', nrays, ' accepted steps:', steps
  call srt_free(ppos); call srt_free(pdir); call srt_free(pw)
  call srt_model_destroy(model)
end program srt_fortran_driver
