! srt_bindc.f90 -- Fortran-callable shim over the C ABI of libsrt_hip.so (include/srt.h).
!
! A Fortran program (e.g. a trimmed raytracer_driver) replaces its serial loop
!     do ... ; call raytracer_run(pos,time,vprel,...,pos0,dir0,w,dt0,...) ; write records ; end do
! (fortran/raytracer_driver.f95:1144-1232) by ONE call of srt_trace_batch over the whole ray file and one
! call of srt_write_ray_file.  Arrays are passed in Fortran's natural layout: pos0(3,nrays) is exactly the
! C layout pos0[nrays][3]; rows(SRT_ROW, slots, nrays) is rows[nrays][slots][SRT_ROW].
module srt_bindc
  use iso_c_binding
  implicit none
  integer, parameter :: SRT_ROW = 20, SRT_MAXSPEC = 4

  type, bind(C) :: srt_params
     real(c_double) :: dt0, dtmax, tmax, maxerr, minalt, del
     integer(c_int32_t) :: maxsteps, root, fixedstep, outputper, first_attempt_policy, refill_threshold, ray_order
  end type srt_params
  ! srt_damping_params of include/srt.h (the MATLAB post-processor matlab/damping/, SURVEY 8f-3)
  type, bind(C) :: srt_damping_params
     integer(c_int32_t) :: dist, mode, nres
     integer(c_int32_t) :: m(8)
     real(c_double) :: Ne_h, kT, tol
  end type srt_damping_params

  interface
     integer(c_int) function srt_init(device) bind(C, name="srt_init")
       import :: c_int
       integer(c_int), value :: device
     end function srt_init
     function srt_last_error() bind(C, name="srt_last_error")
       import :: c_ptr
       type(c_ptr) :: srt_last_error
     end function srt_last_error
     integer(c_int) function srt_model_create_ngo(configfile, yearday, msec, model) bind(C, name="srt_model_create_ngo")
       import :: c_int, c_char, c_ptr
       character(kind=c_char), dimension(*) :: configfile
       integer(c_int), value :: yearday, msec
       type(c_ptr) :: model
     end function srt_model_create_ngo
     integer(c_int) function srt_model_create_interp_file(gridfile, yearday, msec, model) &
          bind(C, name="srt_model_create_interp_file")
       import :: c_int, c_char, c_ptr
       character(kind=c_char), dimension(*) :: gridfile
       integer(c_int), value :: yearday, msec
       type(c_ptr) :: model
     end function srt_model_create_interp_file
     integer(c_int) function srt_model_create_scattered_file(ptsfile, yearday, msec, window_scale, order, exact, &
          local_window_scale, model) bind(C, name="srt_model_create_scattered_file")
       import :: c_int, c_char, c_ptr, c_double
       character(kind=c_char), dimension(*) :: ptsfile
       integer(c_int), value :: yearday, msec, order, exact
       real(c_double), value :: window_scale, local_window_scale
       type(c_ptr) :: model
     end function srt_model_create_scattered_file
     subroutine srt_model_destroy(model) bind(C, name="srt_model_destroy")
       import :: c_ptr
       type(c_ptr), value :: model
     end subroutine srt_model_destroy
     ! --use_igrf / --use_tsyganenko of the driver (interp_dens_model_adapter.f95:214-267 and twins); coeff_file may be
     ! c_null_char-terminated empty string's address replaced by c_null_ptr semantics: pass c_null_char for the default
     integer(c_int) function srt_model_set_field(model, use_igrf, use_tsyganenko, igrf_coeff_file) &
          bind(C, name="srt_model_set_field")
       import :: c_int, c_ptr
       type(c_ptr), value :: model
       integer(c_int), value :: use_igrf, use_tsyganenko
       type(c_ptr), value :: igrf_coeff_file   ! c_null_ptr = the table shipped beside the library
     end function srt_model_set_field
     integer(c_int) function srt_model_set_tsyganenko_params(model, parmod) bind(C, name="srt_model_set_tsyganenko_params")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: model
       real(c_double), intent(in) :: parmod(10)   ! Pdyn, Dst, ByIMF, BzIMF, W1..W6
     end function srt_model_set_tsyganenko_params
     ! hot-plasma damping along the kept rows: rate(slots,nrays), magnitude(slots,nrays), flag(slots,nrays)
     integer(c_int) function srt_damping(dp, nspec, qs, ms, slots, outputper, nrays, rows, nrows, w0, rate, magnitude, flag) &
          bind(C, name="srt_damping")
       import :: c_int, c_int32_t, c_int64_t, c_double, srt_damping_params
       type(srt_damping_params), intent(in) :: dp
       integer(c_int), value :: nspec
       real(c_double), intent(in) :: qs(*), ms(*)
       integer(c_int32_t), value :: slots, outputper
       integer(c_int64_t), value :: nrays
       real(c_double), intent(in) :: rows(*), w0(*)
       integer(c_int32_t), intent(in) :: nrows(*)
       real(c_double) :: rate(*), magnitude(*)
       integer(c_int32_t) :: flag(*)
     end function srt_damping
     integer(c_int) function srt_model_nspec(model) bind(C, name="srt_model_nspec")
       import :: c_int, c_ptr
       type(c_ptr), value :: model
     end function srt_model_nspec
     integer(c_int) function srt_model_species(model, qs, ms) bind(C, name="srt_model_species")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: model
       real(c_double) :: qs(*), ms(*)
     end function srt_model_species
     ! funcPlasmaParams, batched: x(3,n) -> qs,Ns,ms,nus(4,n), B0(3,n)
     integer(c_int) function srt_plasma_params(model, n, x, qs, Ns, ms, nus, B0) bind(C, name="srt_plasma_params")
       import :: c_int, c_ptr, c_double, c_int64_t
       type(c_ptr), value :: model
       integer(c_int64_t), value :: n
       real(c_double) :: x(3,*), qs(4,*), Ns(4,*), ms(4,*), nus(4,*), B0(3,*)
     end function srt_plasma_params
     integer(c_int32_t) function srt_rows_per_ray(p) bind(C, name="srt_rows_per_ray")
       import :: c_int32_t, srt_params
       type(srt_params) :: p
     end function srt_rows_per_ray
     ! the batched raytracer_run
     integer(c_int) function srt_trace_batch(model, p, nrays, pos0, dir0, w0, rows, nrows, stopcond, accepted_steps) &
          bind(C, name="srt_trace_batch")
       import :: c_int, c_ptr, c_double, c_int64_t, c_int32_t, srt_params
       type(c_ptr), value :: model
       type(srt_params) :: p
       integer(c_int64_t), value :: nrays
       real(c_double) :: pos0(3,*), dir0(3,*), w0(*), rows(*)
       integer(c_int32_t) :: nrows(*), stopcond(*)
       integer(c_int64_t) :: accepted_steps
     end function srt_trace_batch
     integer(c_int64_t) function srt_read_rays_file(path, pos0, dir0, w0) bind(C, name="srt_read_rays_file")
       import :: c_int64_t, c_char, c_ptr
       character(kind=c_char), dimension(*) :: path
       type(c_ptr) :: pos0, dir0, w0
     end function srt_read_rays_file
     integer(c_int) function srt_write_ray_file(path, append, raynum0, nrays, p, nspec, qs, ms, w0, rows, nrows, &
          stopcond) bind(C, name="srt_write_ray_file")
       import :: c_int, c_char, c_double, c_int64_t, c_int32_t, srt_params
       character(kind=c_char), dimension(*) :: path
       integer(c_int), value :: append, nspec
       integer(c_int64_t), value :: raynum0, nrays
       type(srt_params) :: p
       real(c_double) :: qs(*), ms(*), w0(*), rows(*)
       integer(c_int32_t) :: nrows(*), stopcond(*)
     end function srt_write_ray_file
     subroutine srt_free(ptr) bind(C, name="srt_free")
       import :: c_ptr
       type(c_ptr), value :: ptr
     end subroutine srt_free
  end interface
end module srt_bindc
