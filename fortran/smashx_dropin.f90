!  smashx_dropin.f90 -- ISO_C_BINDING shim that makes libsmashx a drop-in for the reference's differentiated
!  core.  It defines the two EXTERNAL procedures the reference's wrapped boundary forwards to through an
!  implicit interface,
!
!      base_forward      (replaces smash/solver/forward/forward.f90:1-80)
!      base_forward_b    (replaces smash/solver/forward/forward_db.f90:10648-10936)
!
!  with the same argument lists, so mw_forward::forward / forward_b (mw_forward.f90:18-68), and every caller
!  above them -- mw_optimize::optimize_lbfgsb (mw_optimize.f90:567,610,670), the f90wrap Python API -- run on
!  the GPU unchanged.  Compiled against the reference's own derived-type modules (read in place, never
!  copied) by oracle/ref/build_ref.sh into oracle/_ref/libsmash_dropin.so.  This file is ours.
!
!  What the shim does: marshals the allocatable components of the derived types to the flat C ABI of
!  include/smashx.h with c_loc, converts the 1-based mesh%path / mesh%gauge_pos to 0-based copies, keeps one
!  plan (routing schedule + HBM-resident forcing) alive across calls for as long as the same forcing arrays
!  are passed, and copies results back into output%qsim / output%cost* / output%fstates / *_b.
!  The reference has no error channel; on a libsmashx error the shim prints the message and stops.

module smashx_c

    use iso_c_binding

    implicit none

    integer(c_int), parameter :: SX_GNP = 16, SX_GNS = 8

    type, bind(C) :: smashx_config
        integer(c_int) :: structure, nrow, ncol, nt, ng
        real(c_float) :: dt, dx
        integer(c_int) :: chunk_steps, pipe_steps, group_size, device
        integer(c_int) :: tile(4)
    end type smashx_config

    type, bind(C) :: smashx_mesh
        type(c_ptr) :: flwdir, flwacc, active_cell, path, gauge_pos, area
        type(c_ptr) :: owner_mask = c_null_ptr
    end type smashx_mesh

    type, bind(C) :: smashx_options
        integer(c_int) :: denormalize_forward, optimize_start_step, njf
        integer(c_int) :: jobs_fun(8)
        real(c_float) :: wjobs_fun(8)
        integer(c_int) :: njr
        integer(c_int) :: jreg_fun(4)
        real(c_float) :: wjreg_fun(4)
        real(c_float) :: wjreg
        integer(c_int) :: optim_parameters(SX_GNP), optim_states(SX_GNS)
        real(c_float) :: lb_parameters(SX_GNP), ub_parameters(SX_GNP), lb_states(SX_GNS), ub_states(SX_GNS)
        type(c_ptr) :: wgauge
    end type smashx_options

    type, bind(C) :: smashx_parameters
        type(c_ptr) :: f(SX_GNP)
    end type smashx_parameters

    type, bind(C) :: smashx_states
        type(c_ptr) :: f(SX_GNS)
    end type smashx_states

    type, bind(C) :: smashx_costs
        real(c_float) :: cost, cost_jobs, cost_jreg
    end type smashx_costs

    interface
        function smashx_last_error() bind(C, name="smashx_last_error") result(p)
            import :: c_ptr
            type(c_ptr) :: p
        end function
        function smashx_plan_create(cfg, mesh, plan) bind(C, name="smashx_plan_create") result(rc)
            import :: c_int, c_ptr, smashx_config, smashx_mesh
            type(smashx_config), intent(in) :: cfg
            type(smashx_mesh), intent(in) :: mesh
            type(c_ptr), intent(out) :: plan
            integer(c_int) :: rc
        end function
        function smashx_plan_destroy(plan) bind(C, name="smashx_plan_destroy") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan
            integer(c_int) :: rc
        end function
        function smashx_set_forcing(plan, prcp, pet, sparse) bind(C, name="smashx_set_forcing") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan, prcp, pet
            integer(c_int), value :: sparse
            integer(c_int) :: rc
        end function
        function smashx_set_qobs(plan, qobs) bind(C, name="smashx_set_qobs") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan, qobs
            integer(c_int) :: rc
        end function
        function smashx_set_options(plan, opt) bind(C, name="smashx_set_options") result(rc)
            import :: c_int, c_ptr, smashx_options
            type(c_ptr), value :: plan
            type(smashx_options), intent(in) :: opt
            integer(c_int) :: rc
        end function
        function smashx_abi_sizes(sizes) bind(C, name="smashx_abi_sizes") result(ver)
            import
            integer(c_int) :: sizes(7)
            integer(c_int) :: ver
        end function smashx_abi_sizes

        function smashx_set_domain_outputs(plan, qsim_domain, net_prcp_domain, sparse) &
        & bind(C, name="smashx_set_domain_outputs") result(rc)
            import
            type(c_ptr), value :: plan, qsim_domain, net_prcp_domain
            integer(c_int), value :: sparse
            integer(c_int) :: rc
        end function smashx_set_domain_outputs

        function smashx_forward_d(plan, params, params_d, params_bgd, states, states_d, states_bgd, qsim, qsim_d, costs, &
        & cost_d) bind(C, name="smashx_forward_d") result(rc)
            import
            type(c_ptr), value :: plan, qsim, qsim_d
            type(smashx_parameters) :: params, params_d, params_bgd
            type(smashx_states) :: states, states_d, states_bgd
            type(smashx_costs) :: costs
            real(c_float) :: cost_d
            integer(c_int) :: rc
        end function smashx_forward_d

        function smashx_forward(plan, params, params_bgd, states, states_bgd, qsim, costs, fstates) &
        & bind(C, name="smashx_forward") result(rc)
            import :: c_int, c_ptr, smashx_parameters, smashx_states, smashx_costs
            type(c_ptr), value :: plan, qsim
            type(smashx_parameters) :: params, params_bgd
            type(smashx_states) :: states, states_bgd, fstates
            type(smashx_costs) :: costs
            integer(c_int) :: rc
        end function
        function smashx_forward_b(plan, params, params_bgd, states, states_bgd, cost_b, qsim, costs, params_b, states_b) &
        & bind(C, name="smashx_forward_b") result(rc)
            import :: c_int, c_float, c_ptr, smashx_parameters, smashx_states, smashx_costs
            type(c_ptr), value :: plan, qsim
            type(smashx_parameters) :: params, params_bgd, params_b
            type(smashx_states) :: states, states_bgd, states_b
            real(c_float), value :: cost_b
            type(smashx_costs) :: costs
            integer(c_int) :: rc
        end function
    end interface

    !  include/smashx.h smashx_forcing_layout: lossless compact residency of the forcing
    type, bind(C) :: smashx_forcing_layout
        integer(c_int) :: compact
        real(c_float) :: prcp_factor
        real(c_float) :: pet_ratio(24)
        integer(c_int) :: pet_hour0
    end type smashx_forcing_layout

    interface
        function smashx_set_forcing_layout(plan, layout) bind(C, name="smashx_set_forcing_layout") result(rc)
            import
            type(c_ptr), value :: plan
            type(smashx_forcing_layout) :: layout
            integer(c_int) :: rc
        end function smashx_set_forcing_layout
    end interface

    !  hourly share of the daily PET in the reference's reader (smash/core/_constant.py:47-75)
    real(c_float), parameter :: sx_ratio_pet_hourly(24) = [0._c_float, 0._c_float, 0._c_float, 0._c_float, 0._c_float, 0._c_float, &
    & 0._c_float, 0.035_c_float, 0.062_c_float, 0.079_c_float, 0.097_c_float, 0.11_c_float, 0.117_c_float, 0.117_c_float, &
    & 0.11_c_float, 0.097_c_float, 0.079_c_float, 0.062_c_float, 0.035_c_float, 0._c_float, 0._c_float, 0._c_float, 0._c_float, &
    & 0._c_float]

    !  one cached plan (the reference calls forward/forward_b many times on the same setup/mesh/input_data)
    type(c_ptr), save :: sx_plan = c_null_ptr
    logical, save :: sx_abi_checked = .false.
    type(c_ptr), save :: sx_key_forcing = c_null_ptr
    integer, save :: sx_key(6) = 0
    !  what the cached plan was built from, beyond the sizes: dt, dx, a hash of the mesh arrays, and a strided hash of the
    !  forcing that is recomputed on every call (in-place writes through the f90wrap setters, or a new Model whose arrays land
    !  at the freed address with the same shape, must not be served the old forcing or mesh)
    real(c_float), save :: sx_key_dt = -1._c_float, sx_key_dx = -1._c_float
    integer(c_long), save :: sx_key_mesh = -1_c_long, sx_key_force = -1_c_long
    integer, save :: sx_key_layout = -1          ! compact forcing requested (hour0 + 1) or not (0)

contains

    !  hour of setup%start_time ("%Y%m%d%H%M", mwd_setup.f90:11, or "YYYY-MM-DD HH:MM" as the user guide writes it): digits 9-10
    integer function sx_start_hour(stamp) result(hh)
        character(len=*), intent(in) :: stamp
        integer :: i, nd, d(12)
        nd = 0
        do i = 1, len_trim(stamp)
            if (stamp(i:i) .ge. "0" .and. stamp(i:i) .le. "9" .and. nd .lt. 12) then
                nd = nd + 1
                d(nd) = iachar(stamp(i:i)) - iachar("0")
            end if
        end do
        hh = 0
        if (nd .ge. 10) hh = mod(10*d(9) + d(10), 24)
    end function sx_start_hour

    !  order-sensitive hash of n 32-bit words taken every `step` (h stays below 2**31, the products below 2**51: no overflow)
    function sx_hash_words(a, n, step, h0) result(h)
        integer(c_int), intent(in) :: a(*)
        integer(c_long), intent(in) :: n, step, h0
        integer(c_long) :: h, i
        h = h0
        do i = 1, n, step
            h = mod(h*1000003_c_long + iand(int(a(i), c_long), 4294967295_c_long) + i, 2147483647_c_long)
        end do
    end function sx_hash_words

    subroutine sx_check(rc, where)
        integer(c_int), intent(in) :: rc
        character(len=*), intent(in) :: where
        character(kind=c_char), pointer :: msg(:)
        integer :: i
        if (rc .ne. 0) then
            call c_f_pointer(smashx_last_error(), msg, [512])
            write (*, '(a,a,a,i0,a)', advance='no') "smashx: ", where, " failed (", rc, "): "
            do i = 1, 512
                if (msg(i) .eq. c_null_char) exit
                write (*, '(a)', advance='no') msg(i)
            end do
            write (*, *)
            error stop 1
        end if
    end subroutine sx_check

    integer function sx_structure_id(structure) result(id)
        character(len=*), intent(in) :: structure
        select case (trim(structure))
        case ("gr-a"); id = 1
        case ("gr-b"); id = 2
        case ("gr-c"); id = 3
        case ("gr-d"); id = 4
        case ("vic-a"); id = 5
        case default; id = 0
        end select
    end function sx_structure_id

    integer function sx_jobs_id(name) result(id)
        character(len=*), intent(in) :: name
        select case (trim(name))
        case ("nse"); id = 1
        case ("kge"); id = 2
        case ("kge2"); id = 3
        case ("se"); id = 4
        case ("rmse"); id = 5
        case ("logarithmic"); id = 6
        case default; id = 99     !  signatures: rejected by smashx_set_options
        end select
    end function sx_jobs_id

    integer function sx_jreg_id(name) result(id)
        character(len=*), intent(in) :: name
        select case (trim(name))
        case ("prior"); id = 1
        case ("smoothing"); id = 2
        case ("hard_smoothing"); id = 3
        case default; id = 99
        end select
    end function sx_jreg_id

end module smashx_c

!  plan management + option transfer shared by the two entry points
module smashx_glue

    use iso_c_binding
    use md_constant
    use mwd_setup
    use mwd_mesh
    use mwd_input_data
    use mwd_parameters
    use mwd_states
    use smashx_c

    implicit none

contains

!  plain = .true. (the hyper entry points): no denormalisation, no regulariser -- base_hyper_forward ignores both
subroutine smashx_prepare(setup, mesh, input_data, plain)

    implicit none

    type(SetupDT), intent(in), target :: setup
    type(MeshDT), intent(in), target :: mesh
    type(Input_DataDT), intent(in), target :: input_data
    logical, intent(in), optional :: plain

    type(smashx_config) :: cfg
    type(smashx_mesh) :: cm
    type(smashx_options) :: opt
    integer(c_int), allocatable, target, save :: path0(:, :), gpos0(:, :)
    real(c_float), allocatable, target, save :: wg(:)
    type(c_ptr) :: fkey
    integer :: key(6), j, want_layout
    type(smashx_forcing_layout) :: lay
    integer(c_long) :: n2, nf, hmesh, hforce, hstep
    integer(c_long), save :: rehash_count = 0
    character(len=8) :: envbuf
    integer :: envstat
    integer(c_int), pointer :: wp(:), we(:)
    integer(c_int) :: abi(7), ver
    type(smashx_parameters) :: abi_p
    type(smashx_states) :: abi_s
    type(smashx_costs) :: abi_c

    !  the derived types above mirror include/smashx.h by hand: refuse to run against a library built from another layout
    if (.not. sx_abi_checked) then
        ver = smashx_abi_sizes(abi)
        if (abi(1) .ne. int(c_sizeof(cfg)) .or. abi(2) .ne. int(c_sizeof(cm)) .or. abi(3) .ne. int(c_sizeof(opt)) .or. &
        &   abi(4) .ne. int(c_sizeof(abi_p)) .or. abi(5) .ne. int(c_sizeof(abi_s)) .or. abi(6) .ne. int(c_sizeof(abi_c))) then
            write (*, '(a,i0,a)') "smashx drop-in: libsmashx (ABI version ", ver, &
            & ") does not match the struct layout this shim was compiled for -- rebuild fortran/smashx_dropin.f90"
            error stop 1
        end if
        sx_abi_checked = .true.
    end if

    if (setup%sparse_storage) then
        fkey = c_loc(input_data%sparse_prcp)
    else
        fkey = c_loc(input_data%prcp)
    end if
    key = [sx_structure_id(setup%structure), mesh%nrow, mesh%ncol, setup%ntime_step, mesh%ng, merge(1, 0, setup%sparse_storage)]
    n2 = int(mesh%nrow, c_long)*int(mesh%ncol, c_long)
    hmesh = sx_hash_words(mesh%flwdir, n2, 1_c_long, 7_c_long)
    hmesh = sx_hash_words(mesh%active_cell, n2, 1_c_long, hmesh)
    hmesh = sx_hash_words(mesh%flwacc, n2, 1_c_long, hmesh)
    if (mesh%ng .gt. 0) then
        hmesh = sx_hash_words(mesh%gauge_pos, 2_c_long*mesh%ng, 1_c_long, hmesh)
        hmesh = sx_hash_words(transfer(mesh%area, [1_c_int]), int(mesh%ng, c_long), 1_c_long, hmesh)
    end if
    !  forcing: fields up to 2**24 values are hashed whole, larger ones by ~65 k samples per field (0.1 ms per call; hashing 70 GB would
    !  cost more than the sweep) -- an in-place edit of a small window of a LARGE field can therefore go unnoticed: set the environment
    !  variable SMASHX_DROPIN_REHASH=1 to force a fresh upload on every call, or touch a sampled value.  transfer() would copy the
    !  arrays, so the words are read in place
    if (setup%sparse_storage) then
        nf = int(size(input_data%sparse_prcp, kind=c_long), c_long)
        call c_f_pointer(c_loc(input_data%sparse_prcp), wp, [nf])
        call c_f_pointer(c_loc(input_data%sparse_pet), we, [nf])
    else
        nf = int(size(input_data%prcp, kind=c_long), c_long)
        call c_f_pointer(c_loc(input_data%prcp), wp, [nf])
        call c_f_pointer(c_loc(input_data%pet), we, [nf])
    end if
    hstep = 1_c_long
    if (nf .gt. 16777216_c_long) hstep = max(1_c_long, nf/65536_c_long)
    hforce = sx_hash_words(wp, nf, hstep, 11_c_long)
    hforce = sx_hash_words(we, nf, hstep, hforce)
    call get_environment_variable("SMASHX_DROPIN_REHASH", envbuf, status=envstat)
    if (envstat .eq. 0 .and. envbuf(1:1) .eq. '1') then
        rehash_count = rehash_count + 1_c_long
        hforce = hforce + rehash_count        ! never equal to the cached value: the forcing is sent again
    end if

    want_layout = 0
    if (setup%daily_interannual_pet .and. abs(setup%dt - 3600._sp) .lt. 0.5_sp .and. setup%prcp_conversion_factor .gt. 0._sp) &
    &   want_layout = 1 + mod(sx_start_hour(setup%start_time) + 1, 24)
    if (want_layout .ne. sx_key_layout .and. c_associated(sx_plan)) then
        call sx_check(smashx_plan_destroy(sx_plan), "plan_destroy")
        sx_plan = c_null_ptr
    end if

    if (c_associated(sx_plan) .and. c_associated(fkey, sx_key_forcing) .and. all(key .eq. sx_key) .and. setup%dt .eq. sx_key_dt &
    &   .and. mesh%dx .eq. sx_key_dx .and. hmesh .eq. sx_key_mesh .and. hforce .ne. sx_key_force) then
        !  same plan, new forcing values (written in place): only the forcing goes up again
        if (setup%sparse_storage) then
            call sx_check(smashx_set_forcing(sx_plan, c_loc(input_data%sparse_prcp), c_loc(input_data%sparse_pet), 1_c_int), &
            & "set_forcing")
        else
            call sx_check(smashx_set_forcing(sx_plan, c_loc(input_data%prcp), c_loc(input_data%pet), 0_c_int), "set_forcing")
        end if
        sx_key_force = hforce
    end if

    if (.not. c_associated(sx_plan) .or. .not. c_associated(fkey, sx_key_forcing) .or. any(key .ne. sx_key) .or. &
    &   setup%dt .ne. sx_key_dt .or. mesh%dx .ne. sx_key_dx .or. hmesh .ne. sx_key_mesh) then
        if (c_associated(sx_plan)) call sx_check(smashx_plan_destroy(sx_plan), "plan_destroy")
        sx_plan = c_null_ptr
        cfg%structure = key(1); cfg%nrow = mesh%nrow; cfg%ncol = mesh%ncol; cfg%nt = setup%ntime_step; cfg%ng = mesh%ng
        cfg%dt = setup%dt; cfg%dx = mesh%dx
        cfg%chunk_steps = 0; cfg%pipe_steps = 0; cfg%group_size = 0; cfg%device = -1
        cfg%tile = 0
        if (allocated(path0)) deallocate (path0)
        if (allocated(gpos0)) deallocate (gpos0)
        allocate (path0(2, mesh%nrow*mesh%ncol), gpos0(max(mesh%ng, 1), 2))
        path0 = mesh%path - 1
        if (mesh%ng .gt. 0) gpos0(1:mesh%ng, :) = mesh%gauge_pos - 1
        cm%flwdir = c_loc(mesh%flwdir); cm%flwacc = c_loc(mesh%flwacc); cm%active_cell = c_loc(mesh%active_cell)
        cm%path = c_loc(path0); cm%gauge_pos = c_loc(gpos0)
        cm%area = c_null_ptr
        if (mesh%ng .gt. 0) cm%area = c_loc(mesh%area)
        call sx_check(smashx_plan_create(cfg, cm, sx_plan), "plan_create")
        !  A Model whose PET came from the daily inter-annual reader (setup%daily_interannual_pet, hourly steps) holds forcing the
        !  plan can keep in 2.17 instead of 8 bytes per cell-step: rain counts x prcp_conversion_factor, daily PET x RATIO_PET_HOURLY.
        !  The library verifies every value bit for bit and keeps fp32 rows by itself when the data is not of that form.
        if (want_layout .gt. 0) then
            lay%compact = 1
            lay%prcp_factor = setup%prcp_conversion_factor
            lay%pet_ratio = sx_ratio_pet_hourly
            lay%pet_hour0 = want_layout - 1                                   ! the first step is start_time + dt
            call sx_check(smashx_set_forcing_layout(sx_plan, lay), "set_forcing_layout")
        end if
        if (setup%sparse_storage) then
            call sx_check(smashx_set_forcing(sx_plan, c_loc(input_data%sparse_prcp), c_loc(input_data%sparse_pet), 1_c_int), &
            & "set_forcing")
        else
            call sx_check(smashx_set_forcing(sx_plan, c_loc(input_data%prcp), c_loc(input_data%pet), 0_c_int), "set_forcing")
        end if
        sx_key_forcing = fkey
        sx_key = key
        sx_key_dt = setup%dt; sx_key_dx = mesh%dx; sx_key_mesh = hmesh; sx_key_force = hforce; sx_key_layout = want_layout
    end if

    if (mesh%ng .gt. 0) call sx_check(smashx_set_qobs(sx_plan, c_loc(input_data%qobs)), "set_qobs")

    opt%denormalize_forward = merge(1, 0, setup%optimize%denormalize_forward)
    opt%optimize_start_step = setup%optimize%optimize_start_step
    opt%njf = setup%optimize%njf
    opt%jobs_fun = 0; opt%wjobs_fun = 0._c_float
    do j = 1, min(setup%optimize%njf, 8)
        opt%jobs_fun(j) = sx_jobs_id(setup%optimize%jobs_fun(j))
        opt%wjobs_fun(j) = setup%optimize%wjobs_fun(j)
    end do
    opt%njr = setup%optimize%njr
    opt%jreg_fun = 0; opt%wjreg_fun = 0._c_float
    do j = 1, min(setup%optimize%njr, 4)
        opt%jreg_fun(j) = sx_jreg_id(setup%optimize%jreg_fun(j))
        opt%wjreg_fun(j) = setup%optimize%wjreg_fun(j)
    end do
    opt%wjreg = setup%optimize%wjreg
    opt%optim_parameters = setup%optimize%optim_parameters
    opt%optim_states = setup%optimize%optim_states
    opt%lb_parameters = setup%optimize%lb_parameters
    opt%ub_parameters = setup%optimize%ub_parameters
    opt%lb_states = setup%optimize%lb_states
    opt%ub_states = setup%optimize%ub_states
    if (allocated(wg)) deallocate (wg)
    allocate (wg(max(mesh%ng, 1)))
    wg = 0._c_float
    if (mesh%ng .gt. 0) wg(1:mesh%ng) = setup%optimize%wgauge
    opt%wgauge = c_loc(wg)
    if (present(plain)) then
        if (plain) then
            opt%denormalize_forward = 0
            opt%njr = 0
        end if
    end if
    call sx_check(smashx_set_options(sx_plan, opt), "set_options")

end subroutine smashx_prepare

subroutine smashx_pack_parameters(p, c)
    implicit none
    type(ParametersDT), intent(in), target :: p
    type(smashx_parameters), intent(out) :: c
    !  md_constant.f90:37-57 order
    c%f(1) = c_loc(p%ci); c%f(2) = c_loc(p%cp); c%f(3) = c_loc(p%beta); c%f(4) = c_loc(p%cft)
    c%f(5) = c_loc(p%cst); c%f(6) = c_loc(p%alpha); c%f(7) = c_loc(p%exc); c%f(8) = c_loc(p%b)
    c%f(9) = c_loc(p%cusl1); c%f(10) = c_loc(p%cusl2); c%f(11) = c_loc(p%clsl); c%f(12) = c_loc(p%ks)
    c%f(13) = c_loc(p%ds); c%f(14) = c_loc(p%dsm); c%f(15) = c_loc(p%ws); c%f(16) = c_loc(p%lr)
end subroutine smashx_pack_parameters

subroutine smashx_pack_states(s, c)
    implicit none
    type(StatesDT), intent(in), target :: s
    type(smashx_states), intent(out) :: c
    c%f(1) = c_loc(s%hi); c%f(2) = c_loc(s%hp); c%f(3) = c_loc(s%hft); c%f(4) = c_loc(s%hst)
    c%f(5) = c_loc(s%husl1); c%f(6) = c_loc(s%husl2); c%f(7) = c_loc(s%hlsl); c%f(8) = c_loc(s%hlr)
end subroutine smashx_pack_states

end module smashx_glue

subroutine base_forward(setup, mesh, input_data, parameters, parameters_bgd, states, states_bgd, output, cost)

    use iso_c_binding
    use md_constant
    use mwd_setup
    use mwd_mesh
    use mwd_input_data
    use mwd_parameters
    use mwd_states
    use mwd_output
    use smashx_c
    use smashx_glue

    implicit none

    type(SetupDT), intent(in), target :: setup
    type(MeshDT), intent(in), target :: mesh
    type(Input_DataDT), intent(in), target :: input_data
    type(ParametersDT), intent(inout), target :: parameters
    type(ParametersDT), intent(in), target :: parameters_bgd
    type(StatesDT), intent(inout), target :: states
    type(StatesDT), intent(in), target :: states_bgd
    type(OutputDT), intent(inout), target :: output
    real(sp), intent(inout) :: cost

    type(smashx_parameters) :: cp, cpb
    type(smashx_states) :: cs, csb, cf
    type(smashx_costs) :: cc
    type(c_ptr) :: qs, qdom, pdom
    integer(c_int) :: sp_flag

    !$omp critical (smashx_gpu)
    call smashx_prepare(setup, mesh, input_data)
    call smashx_pack_parameters(parameters, cp)
    call smashx_pack_parameters(parameters_bgd, cpb)
    call smashx_pack_states(states, cs)
    call smashx_pack_states(states_bgd, csb)
    call smashx_pack_states(output%fstates, cf)
    qs = c_null_ptr
    if (mesh%ng .gt. 0) qs = c_loc(output%qsim)
    !  optional whole-domain stores (md_forward_structure.f90:158-194)
    qdom = c_null_ptr
    pdom = c_null_ptr
    sp_flag = 0
    if (setup%sparse_storage) sp_flag = 1
    if (setup%save_qsim_domain) then
        if (setup%sparse_storage) then
            qdom = c_loc(output%sparse_qsim_domain)
        else
            qdom = c_loc(output%qsim_domain)
        end if
    end if
    if (setup%save_net_prcp_domain) then
        if (setup%sparse_storage) then
            pdom = c_loc(output%sparse_net_prcp_domain)
        else
            pdom = c_loc(output%net_prcp_domain)
        end if
    end if
    call sx_check(smashx_set_domain_outputs(sx_plan, qdom, pdom, sp_flag), "set_domain_outputs")
    call sx_check(smashx_forward(sx_plan, cp, cpb, cs, csb, qs, cc, cf), "forward")
    call sx_check(smashx_set_domain_outputs(sx_plan, c_null_ptr, c_null_ptr, sp_flag), "set_domain_outputs")
    cost = cc%cost
    output%cost = cc%cost
    output%cost_jobs = cc%cost_jobs
    output%cost_jreg = cc%cost_jreg
    !$omp end critical (smashx_gpu)

end subroutine base_forward

subroutine base_forward_b(setup, mesh, input_data, parameters, parameters_b, parameters_bgd, parameters_bgd_b, &
& states, states_b, states_bgd, states_bgd_b, output, output_b, cost, cost_b)

    use iso_c_binding
    use md_constant
    use mwd_setup
    use mwd_mesh
    use mwd_input_data
    use mwd_parameters
    use mwd_states
    use mwd_output
    use smashx_c
    use smashx_glue

    implicit none

    type(SetupDT), intent(in), target :: setup
    type(MeshDT), intent(in), target :: mesh
    type(Input_DataDT), intent(in), target :: input_data
    type(ParametersDT), intent(inout), target :: parameters, parameters_b
    type(ParametersDT), intent(in), target :: parameters_bgd
    type(ParametersDT) :: parameters_bgd_b
    type(StatesDT), intent(inout), target :: states, states_b
    type(StatesDT), intent(in), target :: states_bgd
    type(StatesDT) :: states_bgd_b
    type(OutputDT), intent(inout), target :: output
    type(OutputDT), intent(inout) :: output_b      !  scratch in the reference (OUTPUTDT_DIFF); untouched here
    real(sp), intent(inout) :: cost, cost_b

    type(smashx_parameters) :: cp, cpb, cpg
    type(smashx_states) :: cs, csb, csg
    type(smashx_costs) :: cc
    type(c_ptr) :: qs

    !$omp critical (smashx_gpu)
    call smashx_prepare(setup, mesh, input_data)
    call smashx_pack_parameters(parameters, cp)
    call smashx_pack_parameters(parameters_bgd, cpb)
    call smashx_pack_parameters(parameters_b, cpg)
    call smashx_pack_states(states, cs)
    call smashx_pack_states(states_bgd, csb)
    call smashx_pack_states(states_b, csg)
    qs = c_null_ptr
    if (mesh%ng .gt. 0) qs = c_loc(output%qsim)
    call sx_check(smashx_forward_b(sx_plan, cp, cpb, cs, csb, cost_b, qs, cc, cpg, csg), "forward_b")
    cost = cc%cost
    output%cost = cc%cost
    output%cost_jobs = cc%cost_jobs
    output%cost_jreg = cc%cost_jreg
    !$omp end critical (smashx_gpu)

end subroutine base_forward_b

!  Tangent-linear model: replaces BASE_FORWARD_D (forward_db.f90:10517-10601) behind mw_forward::forward_d
!  (mw_forward.f90:70-97).  parameters_bgd_d / states_bgd_d are passive in the reference; output_d is the one-field
!  OUTPUTDT_DIFF (qsim), which callers hand over as an OutputDT.
subroutine base_forward_d(setup, mesh, input_data, parameters, parameters_d, parameters_bgd, parameters_bgd_d, &
& states, states_d, states_bgd, states_bgd_d, output, output_d, cost, cost_d)

    use iso_c_binding
    use md_constant
    use mwd_setup
    use mwd_mesh
    use mwd_input_data
    use mwd_parameters
    use mwd_states
    use mwd_output
    use smashx_c
    use smashx_glue

    implicit none

    type(SetupDT), intent(in), target :: setup
    type(MeshDT), intent(in), target :: mesh
    type(Input_DataDT), intent(in), target :: input_data
    type(ParametersDT), intent(inout), target :: parameters, parameters_d
    type(ParametersDT), intent(in), target :: parameters_bgd
    type(ParametersDT), intent(in) :: parameters_bgd_d
    type(StatesDT), intent(inout), target :: states, states_d
    type(StatesDT), intent(in), target :: states_bgd
    type(StatesDT), intent(in) :: states_bgd_d
    type(OutputDT), intent(inout), target :: output, output_d
    real(sp), intent(inout) :: cost, cost_d

    type(smashx_parameters) :: cp, cpd, cpb
    type(smashx_states) :: cs, csd, csb
    type(smashx_costs) :: cc
    type(c_ptr) :: qs, qd
    real(c_float) :: cd

    !$omp critical (smashx_gpu)
    call smashx_prepare(setup, mesh, input_data)
    call smashx_pack_parameters(parameters, cp)
    call smashx_pack_parameters(parameters_d, cpd)
    call smashx_pack_parameters(parameters_bgd, cpb)
    call smashx_pack_states(states, cs)
    call smashx_pack_states(states_d, csd)
    call smashx_pack_states(states_bgd, csb)
    qs = c_null_ptr
    qd = c_null_ptr
    if (mesh%ng .gt. 0) then
        qs = c_loc(output%qsim)
        qd = c_loc(output_d%qsim)
    end if
    cd = 0._c_float
    call sx_check(smashx_forward_d(sx_plan, cp, cpd, cpb, cs, csd, csb, qs, qd, cc, cd), "forward_d")
    cost_d = cd
    !  the reference's COMPUTE_COST_D does not assign cost; the drop-in returns the value it has anyway
    cost = cc%cost
    output%cost = cc%cost
    output%cost_jobs = cc%cost_jobs
    output%cost_jreg = cc%cost_jreg
    !$omp end critical (smashx_gpu)

end subroutine base_forward_d

!  The hyper-linear / hyper-polynomial mappings (forward.f90:82-157, BASE_HYPER_FORWARD_B forward_db.f90:11231-11560): the
!  descriptor -> parameter mapping and its adjoint are one pass over the parameter planes and stay the reference's own host
!  code (mwd_parameters_manipulation, mwd_states_manipulation and their _DIFF twins); the time loop, the cost and their
!  adjoints between them run on the GPU.  base_hyper_forward leaves states at their final values and ignores
!  denormalize_forward and the regularisers (hyper_compute_cost: jreg = 0).
subroutine base_hyper_forward(setup, mesh, input_data, parameters, hyper_parameters, hyper_parameters_bgd, &
& states, hyper_states, hyper_states_bgd, output, cost)

    use iso_c_binding
    use md_constant
    use mwd_setup
    use mwd_mesh
    use mwd_input_data
    use mwd_parameters
    use mwd_states
    use mwd_output
    use mwd_parameters_manipulation, only: hyper_parameters_to_parameters
    use mwd_states_manipulation, only: hyper_states_to_states
    use smashx_c
    use smashx_glue

    implicit none

    type(SetupDT), intent(in), target :: setup
    type(MeshDT), intent(in), target :: mesh
    type(Input_DataDT), intent(inout), target :: input_data
    type(ParametersDT), intent(inout), target :: parameters
    type(Hyper_ParametersDT), intent(in) :: hyper_parameters, hyper_parameters_bgd
    type(StatesDT), intent(inout), target :: states
    type(Hyper_StatesDT), intent(in) :: hyper_states, hyper_states_bgd
    type(OutputDT), intent(inout), target :: output
    real(sp), intent(inout) :: cost

    type(smashx_parameters) :: cp
    type(smashx_states) :: cs, cf
    type(smashx_costs) :: cc
    type(c_ptr) :: qs

    !$omp critical (smashx_gpu)
    call hyper_parameters_to_parameters(hyper_parameters, parameters, setup, mesh, input_data)
    call hyper_states_to_states(hyper_states, states, setup, mesh, input_data)
    call smashx_prepare(setup, mesh, input_data, .true.)
    call smashx_pack_parameters(parameters, cp)
    call smashx_pack_states(states, cs)
    call smashx_pack_states(output%fstates, cf)
    qs = c_null_ptr
    if (mesh%ng .gt. 0) qs = c_loc(output%qsim)
    call sx_check(smashx_forward(sx_plan, cp, cp, cs, cs, qs, cc, cf), "hyper forward")
    states = output%fstates                 !  forward.f90:150: fstates = states, no restore
    cost = cc%cost
    output%cost = cc%cost
    output%cost_jobs = cc%cost_jobs
    !$omp end critical (smashx_gpu)

end subroutine base_hyper_forward

subroutine base_hyper_forward_b(setup, mesh, input_data, parameters, parameters_b, hyper_parameters, hyper_parameters_b, &
& hyper_parameters_bgd, states, states_b, hyper_states, hyper_states_b, hyper_states_bgd, output, output_b, cost, cost_b)

    use iso_c_binding
    use md_constant
    use mwd_setup
    use mwd_mesh
    use mwd_input_data
    use mwd_parameters_diff
    use mwd_states_diff
    use mwd_output_diff
    use mwd_parameters_manipulation_diff, only: hyper_parameters_to_parameters, hyper_parameters_to_parameters_b, &
    & set_hyper_parameters
    use mwd_states_manipulation_diff, only: hyper_states_to_states, hyper_states_to_states_b, set_hyper_states
    use smashx_c
    use smashx_glue, only: smashx_prepare, sx_plan, sx_check

    implicit none

    type(SetupDT), intent(in), target :: setup
    type(MeshDT), intent(in), target :: mesh
    type(Input_DataDT), intent(inout), target :: input_data
    type(ParametersDT), intent(inout), target :: parameters, parameters_b
    type(Hyper_ParametersDT), intent(in) :: hyper_parameters, hyper_parameters_bgd
    type(Hyper_ParametersDT), intent(inout) :: hyper_parameters_b
    type(StatesDT), intent(inout), target :: states, states_b
    type(Hyper_StatesDT), intent(in) :: hyper_states, hyper_states_bgd
    type(Hyper_StatesDT), intent(inout) :: hyper_states_b
    type(OutputDT), intent(inout), target :: output
    type(OutputDT), intent(inout) :: output_b
    real(sp), intent(inout) :: cost, cost_b

    type(smashx_parameters) :: cp, cpg
    type(smashx_states) :: cs, csg
    type(smashx_costs) :: cc
    type(c_ptr) :: qs

    !$omp critical (smashx_gpu)
    call hyper_parameters_to_parameters(hyper_parameters, parameters, setup, mesh, input_data)
    call hyper_states_to_states(hyper_states, states, setup, mesh, input_data)
    call smashx_prepare(setup, mesh, input_data, .true.)
    !  the _DIFF twins of the derived types have the layout of the originals: marshal by address
    cp%f(1) = c_loc(parameters%ci); cp%f(2) = c_loc(parameters%cp); cp%f(3) = c_loc(parameters%beta)
    cp%f(4) = c_loc(parameters%cft); cp%f(5) = c_loc(parameters%cst); cp%f(6) = c_loc(parameters%alpha)
    cp%f(7) = c_loc(parameters%exc); cp%f(8) = c_loc(parameters%b); cp%f(9) = c_loc(parameters%cusl1)
    cp%f(10) = c_loc(parameters%cusl2); cp%f(11) = c_loc(parameters%clsl); cp%f(12) = c_loc(parameters%ks)
    cp%f(13) = c_loc(parameters%ds); cp%f(14) = c_loc(parameters%dsm); cp%f(15) = c_loc(parameters%ws)
    cp%f(16) = c_loc(parameters%lr)
    cpg%f(1) = c_loc(parameters_b%ci); cpg%f(2) = c_loc(parameters_b%cp); cpg%f(3) = c_loc(parameters_b%beta)
    cpg%f(4) = c_loc(parameters_b%cft); cpg%f(5) = c_loc(parameters_b%cst); cpg%f(6) = c_loc(parameters_b%alpha)
    cpg%f(7) = c_loc(parameters_b%exc); cpg%f(8) = c_loc(parameters_b%b); cpg%f(9) = c_loc(parameters_b%cusl1)
    cpg%f(10) = c_loc(parameters_b%cusl2); cpg%f(11) = c_loc(parameters_b%clsl); cpg%f(12) = c_loc(parameters_b%ks)
    cpg%f(13) = c_loc(parameters_b%ds); cpg%f(14) = c_loc(parameters_b%dsm); cpg%f(15) = c_loc(parameters_b%ws)
    cpg%f(16) = c_loc(parameters_b%lr)
    cs%f(1) = c_loc(states%hi); cs%f(2) = c_loc(states%hp); cs%f(3) = c_loc(states%hft); cs%f(4) = c_loc(states%hst)
    cs%f(5) = c_loc(states%husl1); cs%f(6) = c_loc(states%husl2); cs%f(7) = c_loc(states%hlsl); cs%f(8) = c_loc(states%hlr)
    csg%f(1) = c_loc(states_b%hi); csg%f(2) = c_loc(states_b%hp); csg%f(3) = c_loc(states_b%hft); csg%f(4) = c_loc(states_b%hst)
    csg%f(5) = c_loc(states_b%husl1); csg%f(6) = c_loc(states_b%husl2); csg%f(7) = c_loc(states_b%hlsl)
    csg%f(8) = c_loc(states_b%hlr)
    qs = c_null_ptr
    if (mesh%ng .gt. 0) qs = c_loc(output%qsim)
    call sx_check(smashx_forward_b(sx_plan, cp, cp, cs, cs, cost_b, qs, cc, cpg, csg), "hyper forward_b")
    cost = cc%cost
    output%cost = cc%cost
    output%cost_jobs = cc%cost_jobs
    !  BASE_HYPER_FORWARD_B: hyper_*_b zeroed (forward_db.f90:11317-11318), then the adjoint of the two mappings (:11557-11560)
    call set_hyper_parameters(setup, hyper_parameters_b, 0._sp)
    call set_hyper_states(setup, hyper_states_b, 0._sp)
    call hyper_states_to_states_b(hyper_states, hyper_states_b, states, states_b, setup, mesh, input_data)
    call hyper_parameters_to_parameters_b(hyper_parameters, hyper_parameters_b, parameters, parameters_b, setup, mesh, input_data)
    !$omp end critical (smashx_gpu)

end subroutine base_hyper_forward_b

!  Tangent of the hyper mappings: replaces BASE_HYPER_FORWARD_D (forward_db.f90:11079-11162) behind mw_forward::hyper_forward_d
!  (mw_forward.f90:154-181).  The tangent of the descriptor -> parameter mapping is the reference's own host code
!  (HYPER_PARAMETERS_TO_PARAMETERS_D, HYPER_STATES_TO_STATES_D); the tangent sweep and the cost tangent run on the GPU.
subroutine base_hyper_forward_d(setup, mesh, input_data, parameters, parameters_d, hyper_parameters, hyper_parameters_d, &
& hyper_parameters_bgd, states, states_d, hyper_states, hyper_states_d, hyper_states_bgd, output, output_d, cost, cost_d)

    use iso_c_binding
    use md_constant
    use mwd_setup
    use mwd_mesh
    use mwd_input_data
    use mwd_parameters_diff
    use mwd_states_diff
    use mwd_output_diff
    use mwd_parameters_manipulation_diff, only: hyper_parameters_to_parameters_d
    use mwd_states_manipulation_diff, only: hyper_states_to_states_d
    use smashx_c
    use smashx_glue, only: smashx_prepare, sx_plan, sx_check

    implicit none

    type(SetupDT), intent(in), target :: setup
    type(MeshDT), intent(in), target :: mesh
    type(Input_DataDT), intent(inout), target :: input_data
    type(ParametersDT), intent(inout), target :: parameters, parameters_d
    type(Hyper_ParametersDT), intent(in) :: hyper_parameters, hyper_parameters_d, hyper_parameters_bgd
    type(StatesDT), intent(inout), target :: states, states_d
    type(Hyper_StatesDT), intent(in) :: hyper_states, hyper_states_d, hyper_states_bgd
    type(OutputDT), intent(inout), target :: output
    type(OutputDT_diff), intent(inout), target :: output_d
    real(sp), intent(inout) :: cost, cost_d

    type(smashx_parameters) :: cp, cpd
    type(smashx_states) :: cs, csd
    type(smashx_costs) :: cc
    type(c_ptr) :: qs, qd
    real(c_float) :: cd

    !$omp critical (smashx_gpu)
    call hyper_parameters_to_parameters_d(hyper_parameters, hyper_parameters_d, parameters, parameters_d, setup, mesh, input_data)
    call hyper_states_to_states_d(hyper_states, hyper_states_d, states, states_d, setup, mesh, input_data)
    call smashx_prepare(setup, mesh, input_data, .true.)
    cp%f(1) = c_loc(parameters%ci); cp%f(2) = c_loc(parameters%cp); cp%f(3) = c_loc(parameters%beta)
    cp%f(4) = c_loc(parameters%cft); cp%f(5) = c_loc(parameters%cst); cp%f(6) = c_loc(parameters%alpha)
    cp%f(7) = c_loc(parameters%exc); cp%f(8) = c_loc(parameters%b); cp%f(9) = c_loc(parameters%cusl1)
    cp%f(10) = c_loc(parameters%cusl2); cp%f(11) = c_loc(parameters%clsl); cp%f(12) = c_loc(parameters%ks)
    cp%f(13) = c_loc(parameters%ds); cp%f(14) = c_loc(parameters%dsm); cp%f(15) = c_loc(parameters%ws)
    cp%f(16) = c_loc(parameters%lr)
    cpd%f(1) = c_loc(parameters_d%ci); cpd%f(2) = c_loc(parameters_d%cp); cpd%f(3) = c_loc(parameters_d%beta)
    cpd%f(4) = c_loc(parameters_d%cft); cpd%f(5) = c_loc(parameters_d%cst); cpd%f(6) = c_loc(parameters_d%alpha)
    cpd%f(7) = c_loc(parameters_d%exc); cpd%f(8) = c_loc(parameters_d%b); cpd%f(9) = c_loc(parameters_d%cusl1)
    cpd%f(10) = c_loc(parameters_d%cusl2); cpd%f(11) = c_loc(parameters_d%clsl); cpd%f(12) = c_loc(parameters_d%ks)
    cpd%f(13) = c_loc(parameters_d%ds); cpd%f(14) = c_loc(parameters_d%dsm); cpd%f(15) = c_loc(parameters_d%ws)
    cpd%f(16) = c_loc(parameters_d%lr)
    cs%f(1) = c_loc(states%hi); cs%f(2) = c_loc(states%hp); cs%f(3) = c_loc(states%hft); cs%f(4) = c_loc(states%hst)
    cs%f(5) = c_loc(states%husl1); cs%f(6) = c_loc(states%husl2); cs%f(7) = c_loc(states%hlsl); cs%f(8) = c_loc(states%hlr)
    csd%f(1) = c_loc(states_d%hi); csd%f(2) = c_loc(states_d%hp); csd%f(3) = c_loc(states_d%hft); csd%f(4) = c_loc(states_d%hst)
    csd%f(5) = c_loc(states_d%husl1); csd%f(6) = c_loc(states_d%husl2); csd%f(7) = c_loc(states_d%hlsl)
    csd%f(8) = c_loc(states_d%hlr)
    qs = c_null_ptr
    qd = c_null_ptr
    if (mesh%ng .gt. 0) then
        qs = c_loc(output%qsim)
        qd = c_loc(output_d%qsim)
    end if
    cd = 0._c_float
    call sx_check(smashx_forward_d(sx_plan, cp, cpd, cp, cs, csd, cs, qs, qd, cc, cd), "hyper forward_d")
    cost_d = cd
    cost = cc%cost
    output%cost = cc%cost
    output%cost_jobs = cc%cost_jobs
    !$omp end critical (smashx_gpu)

end subroutine base_hyper_forward_d
