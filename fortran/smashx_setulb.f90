!  smashx_setulb.f90 -- `setulb` of the reference's lbfgsb.f (smash/solver/optimize/lbfgsb.f:47, called from
!  mw_optimize.f90:580 and :866) on top of the library's own L-BFGS-B (include/smashx.h: smashx_lbfgsb_*, threaded host C++).
!
!  Same argument list and reverse-communication protocol, so the reference's optimize_lbfgsb loop needs no change: link this
!  object instead of lbfgsb.o (oracle/ref/build_ref.sh shows how: the reference's setulb_ is weakened in a copy of its object and
!  this definition takes over).  What the caller reads back is kept: task = 'FG_START' / 'FG_LNSRCH' / 'NEW_X' / 'CONVERGENCE: ...' /
!  'ABNORMAL_TERMINATION_IN_LNSRCH', isave(30) = iterations, isave(34) = evaluations, dsave(13) = |projected gradient|
!  (mw_optimize.f90:625-651).  wa / iwa / csave / lsave are not used: the state lives in the handle, whose address is kept in
!  isave(1:2).  A search the caller ends itself (task = 'STOP: ...' after NEW_X, the normal exit at maxiter) never calls back: its
!  handle is released at the next 'START' (one is remembered), or with the process.
subroutine setulb(n, m, x, l, u, nbd, f, g, factr, pgtol, wa, iwa, task, iprint, csave, lsave, isave, dsave)
    use iso_c_binding
    implicit none
    integer :: n, m, iprint
    integer :: nbd(n), iwa(3*n), isave(44)
    double precision :: f, factr, pgtol
    double precision :: x(n), l(n), u(n), g(n), wa(*), dsave(29)
    character(len=60) :: task, csave
    logical :: lsave(4)

    interface
        function smashx_lbfgsb_create(n, m, lower, upper, factr, pgtol, handle) bind(C, name="smashx_lbfgsb_create") result(rc)
            import :: c_int, c_long, c_double, c_ptr
            integer(c_long), value :: n
            integer(c_int), value :: m
            real(c_double), intent(in) :: lower(*), upper(*)
            real(c_double), value :: factr, pgtol
            type(c_ptr), intent(out) :: handle
            integer(c_int) :: rc
        end function
        function smashx_lbfgsb_step(handle, x, f, g, task) bind(C, name="smashx_lbfgsb_step") result(rc)
            import :: c_int, c_double, c_ptr
            type(c_ptr), value :: handle
            real(c_double) :: x(*)
            real(c_double), value :: f
            real(c_double), intent(in) :: g(*)
            integer(c_int) :: task
            integer(c_int) :: rc
        end function
        function smashx_lbfgsb_destroy(handle) bind(C, name="smashx_lbfgsb_destroy") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: handle
            integer(c_int) :: rc
        end function
        function smashx_lbfgsb_iterations(handle) bind(C, name="smashx_lbfgsb_iterations") result(k)
            import :: c_long, c_ptr
            type(c_ptr), value :: handle
            integer(c_long) :: k
        end function
        function smashx_lbfgsb_evaluations(handle) bind(C, name="smashx_lbfgsb_evaluations") result(k)
            import :: c_long, c_ptr
            type(c_ptr), value :: handle
            integer(c_long) :: k
        end function
        function smashx_lbfgsb_projected_gradient(handle) bind(C, name="smashx_lbfgsb_projected_gradient") result(v)
            import :: c_double, c_ptr
            type(c_ptr), value :: handle
            real(c_double) :: v
        end function
    end interface

    type(c_ptr), save :: remembered = c_null_ptr
    type(c_ptr) :: h
    integer(c_int32_t) :: two(2)
    integer(c_int) :: ctask, rc
    double precision, allocatable :: lo(:), up(:)
    double precision :: big
    integer :: i

    if (task(1:5) .eq. 'START') then
        if (c_associated(remembered)) rc = smashx_lbfgsb_destroy(remembered)
        remembered = c_null_ptr
        big = huge(1.d0)
        big = big + big                          ! +infinity
        allocate (lo(n), up(n))
        do i = 1, n                              ! nbd: 0 unbounded, 1 lower only, 2 both, 3 upper only (lbfgsb.f:60-68)
            lo(i) = -big
            up(i) = big
            if (nbd(i) .eq. 1 .or. nbd(i) .eq. 2) lo(i) = l(i)
            if (nbd(i) .eq. 2 .or. nbd(i) .eq. 3) up(i) = u(i)
        end do
        rc = smashx_lbfgsb_create(int(n, c_long), int(m, c_int), lo, up, factr, pgtol, h)
        deallocate (lo, up)
        if (rc .ne. 0) then
            task = 'ERROR: SMASHX_LBFGSB_CREATE FAILED'
            return
        end if
        remembered = h
        two = transfer(h, two)                   ! the handle's address in two default integers
        isave(1:2) = two
        ctask = 0
    else
        two = isave(1:2)
        h = transfer(two, h)
        if (task(1:2) .eq. 'FG') then
            ctask = 1
        else if (task(1:5) .eq. 'NEW_X') then
            ctask = 2
        else
            return                               ! a stopped search: nothing to do
        end if
    end if

    rc = smashx_lbfgsb_step(h, x, f, g, ctask)
    if (rc .ne. 0) then
        task = 'ERROR: SMASHX_LBFGSB_STEP FAILED'
        return
    end if
    isave(30) = int(smashx_lbfgsb_iterations(h))
    isave(34) = int(smashx_lbfgsb_evaluations(h))
    dsave(13) = smashx_lbfgsb_projected_gradient(h)
    select case (ctask)
    case (1)                                     ! evaluate f and g at x
        if (isave(34) .eq. 0) then
            task = 'FG_START'
        else
            task = 'FG_LNSRCH'
        end if
    case (2)
        task = 'NEW_X'
    case (3)
        if (dsave(13) .le. pgtol) then
            task = 'CONVERGENCE: NORM_OF_PROJECTED_GRADIENT_<=_PGTOL'
        else
            task = 'CONVERGENCE: REL_REDUCTION_OF_F_<=_FACTR*EPSMCH'
        end if
        rc = smashx_lbfgsb_destroy(h)
        remembered = c_null_ptr
    case default
        task = 'ABNORMAL_TERMINATION_IN_LNSRCH'
        rc = smashx_lbfgsb_destroy(h)
        remembered = c_null_ptr
    end select
end subroutine setulb
