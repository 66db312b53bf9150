! ref_harness -- TEST INFRASTRUCTURE ONLY (never linked into, or called by, the product).
!
! Our own program that links against the *reference's* compiled modules (built by
! oracle/build_ref.py from the sources where they lie under /root/reference) and exposes the
! hot-path procedures one layer at a time, so that golden vectors can be captured for
!   G0  funcPlasmaParams            (--mode=params)
!   G1  dispersion_relation / stix_parameters / solve_dispersion_relation   (--mode=disp)
!       is_right_handed              (--mode=rh)
!   G2  dFdk / dFdw / dFdx / raytracer_evalrhs                               (--mode=grad)
!   G3  one rk4 / rk45 step          (--mode=step)
!   G4  raytracer_run over a ray file, written in the driver's .ray record format (--mode=run)
!
! Flags use the reference's own --name=value grammar (fortran/util.f95:53-84 getopt_named) and the
! same names as fortran/raytracer_driver.f95:181-228 and its per-model blocks.
! Inputs are list-directed text (one record per line); outputs are raw float64 streams
! (read with numpy.fromfile) so nothing is lost to decimal formatting; --mode=run additionally
! writes the text .ray file exactly as raytracer_driver.f95:1197-1217 does.
program ref_harness
  use types
  use util
  use constants
  use raytracer
  use ngo_dens_model_adapter, only : fngo=>funcPlasmaParams, ngoStateData, &
       ngoStateDataP, ngosetup=>setup
  use interp_dens_model_adapter, only : finterp=>funcPlasmaParams, &
       interpStateData, interpStateDataP, interpsetup=>setup
  use scattered_interp_dens_model_adapter, only : &
       fscat=>funcPlasmaParams, scatteredinterpStateData, &
       scatteredinterpStateDataP, scatsetup=>setup
  implicit none

  character(len=10000) :: buffer, mode, infile_name, outfile_name, rayfile_name
  character(len=10000) :: configfile
  integer :: foundopt, modelnum, sz
  real(kind=DP) :: tmpinput
  character, allocatable :: data(:)
  integer :: itime(2), use_tsy, use_igrf

  type(ngoStateData), target :: ngo_sd
  type(ngoStateDataP) :: ngo_sdp
  type(interpStateData), target :: int_sd
  type(interpStateDataP) :: int_sdp
  type(scatteredinterpStateData), target :: sc_sd
  type(scatteredinterpStateDataP) :: sc_sdp

  modelnum = 1
  itime = (/ 2010001, 0 /)
  use_tsy = 0
  use_igrf = 0
  mode = ' '
  infile_name = ' '
  outfile_name = ' '
  rayfile_name = ' '

  call getopt_named('mode', mode, foundopt)
  call getopt_named('in', infile_name, foundopt)
  call getopt_named('out', outfile_name, foundopt)
  call getopt_named('rayout', rayfile_name, foundopt)
  call getopt_named('modelnum', buffer, foundopt)
  if (foundopt == 1) then
     read(buffer,*) tmpinput
     modelnum = floor(tmpinput)
  end if
  call getopt_named('yearday', buffer, foundopt)
  if (foundopt == 1) then
     read(buffer,*) tmpinput
     itime(1) = floor(tmpinput)
  end if
  call getopt_named('milliseconds_day', buffer, foundopt)
  if (foundopt == 1) then
     read(buffer,*) tmpinput
     itime(2) = floor(tmpinput)
  end if

  call getopt_named('use_igrf', buffer, foundopt)
  if (foundopt == 1) then
     read(buffer,*) tmpinput
     use_igrf = floor(tmpinput)
  end if
  call getopt_named('use_tsyganenko', buffer, foundopt)
  if (foundopt == 1) then
     read(buffer,*) tmpinput
     use_tsy = floor(tmpinput)
  end if

  if (trim(mode) == 'rh') then
     call do_rh()
     stop
  end if
  if (trim(mode) == 't04') then
     call do_t04()
     stop
  end if
  if (trim(mode) == 'ext') then
     call do_ext()
     stop
  end if

  if (modelnum == 1) then
     call getopt_named('ngo_configfile', configfile, foundopt)
     ngo_sd%itime = itime
     ngo_sd%use_tsyganenko = use_tsy
     ngo_sd%use_igrf = use_igrf
     ngo_sd%Pdyn = 4.0_DP; ngo_sd%Dst = 1.0_DP; ngo_sd%ByIMF = 0.0_DP; ngo_sd%BzIMF = -5.0_DP
     ngo_sd%W1 = 0.132_DP; ngo_sd%W2 = 0.303_DP; ngo_sd%W3 = 0.083_DP
     ngo_sd%W4 = 0.070_DP; ngo_sd%W5 = 0.211_DP; ngo_sd%W6 = 0.308_DP
     ngo_sdp%p => ngo_sd
     sz = size(transfer(ngo_sdp, data))
     allocate(data(sz))
     data = transfer(ngo_sdp, data)
     call ngosetup(ngo_sd, trim(configfile))
     call dispatch(fngo, 1.0e-4_DP)
  else if (modelnum == 3) then
     call getopt_named('interp_interpfile', configfile, foundopt)
     int_sd%itime = itime
     int_sd%use_tsyganenko = use_tsy
     int_sd%use_igrf = use_igrf
     int_sd%Pdyn = 4.0_DP; int_sd%Dst = 1.0_DP; int_sd%ByIMF = 0.0_DP; int_sd%BzIMF = -5.0_DP
     int_sd%W1 = 0.132_DP; int_sd%W2 = 0.303_DP; int_sd%W3 = 0.083_DP
     int_sd%W4 = 0.070_DP; int_sd%W5 = 0.211_DP; int_sd%W6 = 0.308_DP
     int_sdp%p => int_sd
     sz = size(transfer(int_sdp, data))
     allocate(data(sz))
     data = transfer(int_sdp, data)
     call interpsetup(int_sd, trim(configfile))
     call dispatch(finterp, 1.0e-6_DP)
  else if (modelnum == 4) then
     call getopt_named('interp_interpfile', configfile, foundopt)
     sc_sd%itime = itime
     sc_sd%use_tsyganenko = use_tsy
     sc_sd%use_igrf = use_igrf
     sc_sd%Pdyn = 4.0_DP; sc_sd%Dst = 1.0_DP; sc_sd%ByIMF = 0.0_DP; sc_sd%BzIMF = -5.0_DP
     sc_sd%W1 = 0.132_DP; sc_sd%W2 = 0.303_DP; sc_sd%W3 = 0.083_DP
     sc_sd%W4 = 0.070_DP; sc_sd%W5 = 0.211_DP; sc_sd%W6 = 0.308_DP
     sc_sd%window_scale = 1.5_DP
     sc_sd%order = 2
     sc_sd%exact = 0
     sc_sd%scaled = 0
     sc_sd%local_window_scale = 5.0_DP
     call getopt_named('scattered_interp_window_scale', buffer, foundopt)
     if (foundopt == 1) read(buffer,*) sc_sd%window_scale
     call getopt_named('scattered_interp_order', buffer, foundopt)
     if (foundopt == 1) then
        read(buffer,*) tmpinput
        sc_sd%order = floor(tmpinput)
     end if
     call getopt_named('scattered_interp_exact', buffer, foundopt)
     if (foundopt == 1) then
        read(buffer,*) tmpinput
        sc_sd%exact = floor(tmpinput)
     end if
     call getopt_named('scattered_interp_local_window_scale', buffer, foundopt)
     if (foundopt == 1) read(buffer,*) sc_sd%local_window_scale
     sc_sdp%p => sc_sd
     sz = size(transfer(sc_sdp, data))
     allocate(data(sz))
     data = transfer(sc_sdp, data)
     call scatsetup(sc_sd, trim(configfile))
     if (trim(mode) == 'scatroot') then
        ! which sample the reference's kd-tree has as its root (it depends on the compiler's RNG through randperm,
        ! scattered_interp_dens_model_adapter.f95:137-165): position, then the stored values (the last one is the
        ! nearest-sample distance, which the reference leaves at 0 for exactly this sample), then maxnearest
        write(*,'(a,20es26.17e3)') 'SCAT_ROOT ', sc_sd%tree%point, sc_sd%tree%val, sc_sd%maxnearest
        stop
     end if
     call dispatch(fscat, 1.0e-6_DP)
  else
     print *, 'ref_harness: unsupported modelnum'
     stop 2
  end if

contains

  subroutine dispatch(f, del)
    interface
       subroutine f(x, qs, Ns, ms, nus, B0, funcPlasmaParamsData)
         use types
         real(kind=DP) :: x(3)
         real(kind=DP), allocatable :: qs(:), Ns(:), ms(:), nus(:)
         real(kind=DP) :: B0(3)
         character :: funcPlasmaParamsData(:)
       end subroutine f
    end interface
    real(kind=DP) :: del
    select case (trim(mode))
    case ('params')
       call do_params(f)
    case ('disp')
       call do_disp(f)
    case ('grad')
       call do_grad(f)
    case ('step')
       call do_step(f)
    case ('run')
       call do_run(f, del)
    case default
       print *, 'ref_harness: unknown mode ', trim(mode)
       stop 2
    end select
  end subroutine dispatch

  subroutine do_params(f)
    interface
       subroutine f(x, qs, Ns, ms, nus, B0, funcPlasmaParamsData)
         use types
         real(kind=DP) :: x(3)
         real(kind=DP), allocatable :: qs(:), Ns(:), ms(:), nus(:)
         real(kind=DP) :: B0(3)
         character :: funcPlasmaParamsData(:)
       end subroutine f
    end interface
    real(kind=DP) :: x(3), B0(3)
    real(kind=DP), allocatable :: qs(:), Ns(:), ms(:), nus(:)
    integer :: status
    open(unit=71, file=trim(infile_name), status='old')
    open(unit=72, file=trim(outfile_name), access='stream', form='unformatted', status='replace')
    do
       read(71, *, iostat=status) x
       if (status /= 0) exit
       call f(x, qs, Ns, ms, nus, B0, data)
       write(72) qs, Ns, ms, nus, B0
    end do
    close(71)
    close(72)
  end subroutine do_params

  subroutine do_disp(f)
    interface
       subroutine f(x, qs, Ns, ms, nus, B0, funcPlasmaParamsData)
         use types
         real(kind=DP) :: x(3)
         real(kind=DP), allocatable :: qs(:), Ns(:), ms(:), nus(:)
         real(kind=DP) :: B0(3)
         character :: funcPlasmaParamsData(:)
       end subroutine f
    end interface
    real(kind=DP) :: x(3), k(3), w, B0(3), Fv, S, D, P, R, L
    real(kind=DP), allocatable :: qs(:), Ns(:), ms(:), nus(:)
    complex(kind=DP) :: k1, k2
    integer :: status
    open(unit=71, file=trim(infile_name), status='old')
    open(unit=72, file=trim(outfile_name), access='stream', form='unformatted', status='replace')
    do
       read(71, *, iostat=status) x, k, w
       if (status /= 0) exit
       call f(x, qs, Ns, ms, nus, B0, data)
       Fv = dispersion_relation(k*C/w, w, qs, Ns, ms, nus, B0)
       call stix_parameters(w, qs, Ns, ms, nus, sqrt(dot_product(B0,B0)), S, D, P, R, L)
       call solve_dispersion_relation(k, w, x, k1, k2, f, data)
       write(72) Fv, S, D, P, R, L, real(k1), aimag(k1), real(k2), aimag(k2)
    end do
    close(71)
    close(72)
  end subroutine do_disp

  ! T04_s itself (tsyganenko/TS05_aka_TS04.for:5): rows "parmod(10) ps x y z" (GSM, R_E) -> bx by bz (REAL, widened)
  subroutine do_t04()
    real :: parmod(10), ps, x, y, z, bx, by, bz
    integer :: status
    external :: T04_s
    open(unit=71, file=trim(infile_name), status='old')
    open(unit=72, file=trim(outfile_name), access='stream', form='unformatted', status='replace')
    do
       read(71, *, iostat=status) parmod, ps, x, y, z
       if (status /= 0) exit
       call T04_s(0, parmod, ps, x, y, z, bx, by, bz)
       write(72) real(bx, DP), real(by, DP), real(bz, DP)
    end do
    close(71)
    close(72)
  end subroutine do_t04

  ! EXTERN (TS05_aka_TS04.for:118) with its module outputs: first record = the 69 model coefficients (T04_s's DATA A,
  ! supplied by the caller), then rows "pdyn dst byimf bzimf w1..w6 ps x y z" -> 33 doubles: CF, T1, T2, SRC, PRC, R11,
  ! R12, R21, R22, HIMF, total (3 each)
  subroutine do_ext()
    real(kind=DP) :: a(69), pdyn, dst, bximf, byimf, bzimf, w(6), ps, x, y, z, o(33)
    integer :: status
    external :: EXTERN
    open(unit=71, file=trim(infile_name), status='old')
    open(unit=72, file=trim(outfile_name), access='stream', form='unformatted', status='replace')
    read(71, *) a
    bximf = 0.0_DP
    do
       read(71, *, iostat=status) pdyn, dst, byimf, bzimf, w, ps, x, y, z
       if (status /= 0) exit
       o = 0.0_DP
       call EXTERN(0, 0, 0, 0, a, 69, pdyn, dst, bximf, byimf, bzimf, w(1), w(2), w(3), w(4), w(5), w(6), ps, x, y, z, &
            o(1), o(2), o(3), o(4), o(5), o(6), o(7), o(8), o(9), o(10), o(11), o(12), o(13), o(14), o(15), &
            o(16), o(17), o(18), o(19), o(20), o(21), o(22), o(23), o(24), o(25), o(26), o(27), o(28), o(29), o(30), &
            o(31), o(32), o(33))
       write(72) o
    end do
    close(71)
    close(72)
  end subroutine do_ext

  subroutine do_rh()
    real(kind=DP) :: n2, phi, S, D, P, res
    integer :: status
    open(unit=71, file=trim(infile_name), status='old')
    open(unit=72, file=trim(outfile_name), access='stream', form='unformatted', status='replace')
    do
       read(71, *, iostat=status) n2, phi, S, D, P
       if (status /= 0) exit
       res = 0.0_DP
       if (is_right_handed(n2, phi, S, D, P)) res = 1.0_DP
       write(72) res
    end do
    close(71)
    close(72)
  end subroutine do_rh

  subroutine do_grad(f)
    interface
       subroutine f(x, qs, Ns, ms, nus, B0, funcPlasmaParamsData)
         use types
         real(kind=DP) :: x(3)
         real(kind=DP), allocatable :: qs(:), Ns(:), ms(:), nus(:)
         real(kind=DP) :: B0(3)
         character :: funcPlasmaParamsData(:)
       end subroutine f
    end interface
    real(kind=DP) :: x(3), k(3), w, del, dfdk(3), dfdw, dfdx(3), rhs(7), args(7)
    integer :: status
    open(unit=71, file=trim(infile_name), status='old')
    open(unit=72, file=trim(outfile_name), access='stream', form='unformatted', status='replace')
    do
       read(71, *, iostat=status) x, k, w, del
       if (status /= 0) exit
       dfdk = dispersion_relation_dFdk(k, w, x, 1.0e-8_DP, f, data)
       dfdw = dispersion_relation_dFdw(k, w, x, 1.0e-8_DP, f, data)
       dfdx = dispersion_relation_dFdx(k, w, x, del, f, data)
       args(1:3) = x
       args(4:6) = k
       args(7) = w
       rhs = raytracer_evalrhs(0.0_DP, args, del, f, data)
       write(72) dfdk, dfdw, dfdx, rhs
    end do
    close(71)
    close(72)
  end subroutine do_grad

  subroutine do_step(f)
    interface
       subroutine f(x, qs, Ns, ms, nus, B0, funcPlasmaParamsData)
         use types
         real(kind=DP) :: x(3)
         real(kind=DP), allocatable :: qs(:), Ns(:), ms(:), nus(:)
         real(kind=DP) :: B0(3)
         character :: funcPlasmaParamsData(:)
       end subroutine f
    end interface
    real(kind=DP) :: args(7), dt, del, o4(7), o5(7), r4(7)
    integer :: status
    open(unit=71, file=trim(infile_name), status='old')
    open(unit=72, file=trim(outfile_name), access='stream', form='unformatted', status='replace')
    do
       read(71, *, iostat=status) args, dt, del
       if (status /= 0) exit
       r4 = rk4(0.0_DP, args, del, dt, f, data)
       call rk45(0.0_DP, args, del, dt, f, data, o4, o5)
       write(72) r4, o4, o5
    end do
    close(71)
    close(72)
  end subroutine do_step

  ! Mirrors the ray loop of raytracer_driver.f95:1144-1232 (same record layout and edit descriptors),
  ! plus a lossless float64 stream of every row.
  subroutine do_run(f, del_default)
    interface
       subroutine f(x, qs, Ns, ms, nus, B0, funcPlasmaParamsData)
         use types
         real(kind=DP) :: x(3)
         real(kind=DP), allocatable :: qs(:), Ns(:), ms(:), nus(:)
         real(kind=DP) :: B0(3)
         character :: funcPlasmaParamsData(:)
       end subroutine f
    end interface
    real(kind=DP) :: del_default, del
    real(kind=DP) :: pos0(3), w, dir0(3), dt0, dtmax, maxerr, tmax, minalt
    integer :: fixedstep, root, maxsteps, outputper, stopcond, raynum, status, i, j
    real(kind=DP), allocatable :: pos(:,:), time(:), vprel(:,:), vgrel(:,:), &
         n(:,:), B0(:,:), qs(:,:), ms(:,:), Ns(:,:), nus(:,:)
    integer(kind=8) :: c0, c1, crate, nrows
    logical :: textout

    dt0 = 1.0e-3_DP; dtmax = 0.1_DP; maxerr = 5.0e-4_DP; tmax = 1.0_DP
    minalt = 6.4712e6_DP; fixedstep = 0; root = 2; maxsteps = 2000; outputper = 1
    del = del_default
    call getopt_named('dt0', buffer, foundopt)
    if (foundopt == 1) read(buffer,*) dt0
    call getopt_named('dtmax', buffer, foundopt)
    if (foundopt == 1) read(buffer,*) dtmax
    call getopt_named('tmax', buffer, foundopt)
    if (foundopt == 1) read(buffer,*) tmax
    call getopt_named('maxerr', buffer, foundopt)
    if (foundopt == 1) read(buffer,*) maxerr
    call getopt_named('minalt', buffer, foundopt)
    if (foundopt == 1) read(buffer,*) minalt
    call getopt_named('del', buffer, foundopt)
    if (foundopt == 1) read(buffer,*) del
    call getopt_named('root', buffer, foundopt)
    if (foundopt == 1) then
       read(buffer,*) tmpinput
       root = floor(tmpinput)
    end if
    call getopt_named('fixedstep', buffer, foundopt)
    if (foundopt == 1) then
       read(buffer,*) tmpinput
       fixedstep = floor(tmpinput)
    end if
    call getopt_named('maxsteps', buffer, foundopt)
    if (foundopt == 1) then
       read(buffer,*) tmpinput
       maxsteps = floor(tmpinput)
    end if
    call getopt_named('outputper', buffer, foundopt)
    if (foundopt == 1) then
       read(buffer,*) tmpinput
       outputper = floor(tmpinput)
    end if

    textout = (len_trim(rayfile_name) > 0)
    open(unit=71, file=trim(infile_name), status='old')
    open(unit=72, file=trim(outfile_name), access='stream', form='unformatted', status='replace')
    if (textout) open(unit=73, file=trim(rayfile_name), status='replace')
    raynum = 1
    nrows = 0
    call system_clock(c0, crate)
    do
       read(71, *, iostat=status) pos0, dir0, w
       if (status /= 0) exit
       call raytracer_run(pos, time, vprel, vgrel, n, B0, qs, ms, Ns, nus, stopcond, &
            pos0, dir0, w, dt0, dtmax, maxerr, maxsteps, minalt, root, tmax, &
            fixedstep, del, f, data, raytracer_stopconditions)
       nrows = nrows + size(time,1) - 1
       write(72) real(raynum,kind=DP), real(stopcond,kind=DP), real(size(time,1),kind=DP)
       do i = 1, size(time,1)
          write(72) time(i), pos(:,i), vprel(:,i), vgrel(:,i), n(:,i), B0(:,i), &
               qs(:,i), ms(:,i), Ns(:,i), nus(:,i)
       end do
       if (textout) then
          do i = 1, size(time,1), outputper
             write(73, fmt='(i10, i10, 17es24.15e3, i10)', advance='no') &
                  raynum, stopcond, time(i), pos(:,i), vprel(:,i), vgrel(:,i), n(:,i), &
                  B0(:,i), w, size(qs,1)
             do j = 1, size(qs,1)
                write(73, fmt='(es24.15e3)', advance='no') qs(j,i)
             end do
             do j = 1, size(qs,1)
                write(73, fmt='(es24.15e3)', advance='no') ms(j,i)
             end do
             do j = 1, size(qs,1)
                write(73, fmt='(es24.15e3)', advance='no') Ns(j,i)
             end do
             do j = 1, size(qs,1)
                write(73, fmt='(es24.15e3)', advance='no') nus(j,i)
             end do
             write(73, fmt='(a)') ''
          end do
       end if
       deallocate(pos, time, vprel, vgrel, n, B0, qs, ms, Ns, nus)
       raynum = raynum + 1
    end do
    call system_clock(c1)
    close(71)
    close(72)
    if (textout) close(73)
    print '(a,i12,a,es12.4,a)', 'REF_TIMING accepted_steps=', nrows, ' seconds=', &
         real(c1-c0,kind=DP)/real(crate,kind=DP), ' '
  end subroutine do_run

end program ref_harness
