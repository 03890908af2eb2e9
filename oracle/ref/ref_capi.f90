!  ref_capi.f90 -- TEST INFRASTRUCTURE (oracle side), not product code.
!
!  A bind(C) driver over the UNMODIFIED reference solver modules, compiled together with the
!  reference sources (read in place from /root/reference by oracle/ref/build_ref.sh) into
!  oracle/_ref/libsmash_ref.so.  It fills the reference's derived types from flat arrays, calls the
!  reference's wrapped boundary  mw_forward::forward / forward_b  (smash/solver/forward/
!  mw_forward.f90:18-68) and copies the results back out.  It is what generates tests/golden/*.npz
!  (tests/golden/make_golden.py) and what bench.py times as cpu_baseline kind "reference".
!
!  This file is ours; it contains no reference source text.

module ref_capi

    use iso_c_binding
    use md_constant
    use mwd_setup
    use mwd_mesh
    use mwd_input_data
    use mwd_parameters
    use mwd_states
    use mwd_output
    use mwd_parameters_manipulation
    use mwd_states_manipulation
    use mw_sparse_storage
    use mw_forward, only: forward, forward_b, forward_d, hyper_forward, hyper_forward_b, hyper_forward_d
    use mw_optimize, only: optimize_lbfgsb, optimize_sbs

    implicit none

contains

    function jobs_name(code) result(s)
        integer, intent(in) :: code
        character(20) :: s
        select case (code)
        case (1); s = "nse"
        case (2); s = "kge"
        case (3); s = "kge2"
        case (4); s = "se"
        case (5); s = "rmse"
        case (6); s = "logarithmic"
        case default; s = "..."
        end select
    end function jobs_name

    function jreg_name(code) result(s)
        integer, intent(in) :: code
        character(20) :: s
        select case (code)
        case (1); s = "prior"
        case (2); s = "smoothing"
        case (3); s = "hard_smoothing"
        case default; s = "..."
        end select
    end function jreg_name

    !  icfg(1)  structure id: 1 gr-a, 2 gr-b, 3 gr-c, 4 gr-d, 5 vic-a
    !  icfg(2:5) nrow, ncol, nt, ng
    !  icfg(6)  sparse_storage (0/1)         icfg(7)  denormalize_forward (0/1)
    !  icfg(8)  optimize_start_step (1-based) icfg(9)  njf      icfg(10) njr
    !  icfg(11) mode: 0 = forward, 1 = forward_b, 2 = optimize_lbfgsb (mw_optimize.f90:484-676),
    !           3 = forward_d (tangent model, mw_forward.f90:70-97; entry point ref_run_d only)
    !           4 = hyper_forward, 5 = hyper_forward_b, 6 = hyper_forward_d (mw_forward.f90:99-181; entry point
    !           ref_run_hyper only; mode 6 reads the direction from hyper_p_b / hyper_s_b)
    !           7 = optimize_sbs (mw_optimize.f90:53-294), maxiter in icfg(13)
    !  icfg(14) nd (descriptors)   icfg(15) mapping: 1 hyper-linear, 2 hyper-polynomial   icfg(16) reader-form forcing (see below)
    !  icfg(12) nrep (timing repetitions, >=1)     icfg(13) maxiter (mode 2)
    !  rcfg(1) dt  rcfg(2) dx  rcfg(3) wjreg  rcfg(4) cost_b
    !  Arrays are column-major exactly as the reference holds them; path and gauge_pos are 1-based.
    !  params/states are the (nrow,ncol,GNP)/(nrow,ncol,GNS) packings of get_parameters/get_states
    !  (mwd_parameters_manipulation.f90:59, mwd_states_manipulation.f90:58).
    subroutine ref_core(icfg, rcfg, flwdir, flwacc, path, active_cell, gauge_pos, area, &
    & prcp, pet, qobs, params, params_bgd, states, states_bgd, &
    & wgauge, jobs_codes, wjobs, jreg_codes, wjreg_fun, optim_p, optim_s, lbp, ubp, lbs, ubs, &
    & qsim, costs, fstates, params_out, states_out, params_b, states_b, elapsed, &
    & params_d, states_d, params_bgd_d, states_bgd_d, qsim_d, cost_d_out, &
    & descriptor, hyper_p, hyper_s, hyper_p_b, hyper_s_b)

        integer(c_int), intent(in) :: icfg(16)
        real(c_float), intent(in) :: rcfg(4)
        integer(c_int), intent(in) :: flwdir(icfg(2), icfg(3)), flwacc(icfg(2), icfg(3))
        integer(c_int), intent(in) :: path(2, icfg(2)*icfg(3)), active_cell(icfg(2), icfg(3))
        integer(c_int), intent(in) :: gauge_pos(icfg(5), 2)
        real(c_float), intent(in) :: area(icfg(5))
        real(c_float), intent(in) :: prcp(icfg(2), icfg(3), icfg(4)), pet(icfg(2), icfg(3), icfg(4))
        real(c_float), intent(in) :: qobs(icfg(5), icfg(4))
        real(c_float), intent(in) :: params(icfg(2), icfg(3), GNP), params_bgd(icfg(2), icfg(3), GNP)
        real(c_float), intent(in) :: states(icfg(2), icfg(3), GNS), states_bgd(icfg(2), icfg(3), GNS)
        real(c_float), intent(in) :: wgauge(icfg(5))
        integer(c_int), intent(in) :: jobs_codes(*)
        real(c_float), intent(in) :: wjobs(*)
        integer(c_int), intent(in) :: jreg_codes(*)
        real(c_float), intent(in) :: wjreg_fun(*)
        integer(c_int), intent(in) :: optim_p(GNP), optim_s(GNS)
        real(c_float), intent(in) :: lbp(GNP), ubp(GNP), lbs(GNS), ubs(GNS)
        real(c_float), intent(inout) :: qsim(icfg(5), icfg(4))
        real(c_float), intent(inout) :: costs(3)
        real(c_float), intent(inout) :: fstates(icfg(2), icfg(3), GNS)
        real(c_float), intent(inout) :: params_out(icfg(2), icfg(3), GNP), states_out(icfg(2), icfg(3), GNS)
        real(c_float), intent(inout) :: params_b(icfg(2), icfg(3), GNP), states_b(icfg(2), icfg(3), GNS)
        real(c_double), intent(inout) :: elapsed
        real(c_float), intent(in), optional :: params_d(icfg(2), icfg(3), GNP), states_d(icfg(2), icfg(3), GNS)
        real(c_float), intent(in), optional :: params_bgd_d(icfg(2), icfg(3), GNP), states_bgd_d(icfg(2), icfg(3), GNS)
        real(c_float), intent(inout), optional :: qsim_d(icfg(5), icfg(4))
        real(c_float), intent(inout), optional :: cost_d_out
        real(c_float), intent(in), optional :: descriptor(icfg(2), icfg(3), *)
        real(c_float), intent(in), optional :: hyper_p(1 + icfg(15)*icfg(14), 1, GNP), hyper_s(1 + icfg(15)*icfg(14), 1, GNS)
        real(c_float), intent(inout), optional :: hyper_p_b(1 + icfg(15)*icfg(14), 1, GNP), hyper_s_b(1 + icfg(15)*icfg(14), 1, GNS)

        type(Hyper_ParametersDT) :: hp, hp_b, hp_bgd
        type(Hyper_StatesDT) :: hs, hs_b, hs_bgd
        type(ParametersDT) :: p_d
        type(StatesDT) :: s_d
        type(OutputDT) :: output_d
        real(sp) :: cost_d
        type(SetupDT) :: setup
        type(MeshDT) :: mesh
        type(Input_DataDT) :: input_data
        type(ParametersDT) :: p, p_b, p_bgd, p_bgd_b
        type(StatesDT) :: s, s_b, s_bgd, s_bgd_b
        type(OutputDT) :: output, output_b
        real(sp) :: cost, cost_b
        integer :: nrow, ncol, nt, ng, njf, njr, j, t, rep, nrep
        integer(8) :: c0, c1, crate

        nrow = icfg(2); ncol = icfg(3); nt = icfg(4); ng = icfg(5)
        njf = icfg(9); njr = icfg(10); nrep = max(1, icfg(12))

        select case (icfg(1))
        case (1); setup%structure = "gr-a"
        case (2); setup%structure = "gr-b"
        case (3); setup%structure = "gr-c"
        case (4); setup%structure = "gr-d"
        case (5); setup%structure = "vic-a"
        end select
        setup%dt = rcfg(1)
        setup%sparse_storage = (icfg(6) .ne. 0)
        !  icfg(16) = 1: the forcing was formed the way the reference's reader forms it (_read_input_data.py:176-283): rain = raster
        !  count x 0.1, PET = daily x RATIO_PET_HOURLY, run starting at midnight -- the setup fields a Model built from such data carries
        if (icfg(16) .ne. 0) then
            setup%prcp_conversion_factor = 0.1_c_float
            setup%daily_interannual_pet = .true.
            setup%start_time = "201409150000"
        end if
        setup%ntime_step = nt
        call SetupDT_initialise(setup, icfg(14), ng)
        if (icfg(15) .eq. 1) then
            deallocate (setup%optimize%wgauge)
            call Optimize_SetupDT_initialise(setup%optimize, nt, icfg(14), ng, "hyper-linear", 0, 0)
        else if (icfg(15) .eq. 2) then
            deallocate (setup%optimize%wgauge)
            call Optimize_SetupDT_initialise(setup%optimize, nt, icfg(14), ng, "hyper-polynomial", 0, 0)
        end if

        call MeshDT_initialise(mesh, setup, nrow, ncol, ng)
        mesh%dx = rcfg(2)
        mesh%flwdir = flwdir
        mesh%flwacc = flwacc
        mesh%path = path
        mesh%active_cell = active_cell
        mesh%nac = count(active_cell .eq. 1)
        if (ng .gt. 0) then
            mesh%gauge_pos = gauge_pos
            mesh%area = area
        end if
        if (setup%sparse_storage) call compute_rowcol_to_ind_sparse(mesh)

        call Input_DataDT_initialise(input_data, setup, mesh)
        if (ng .gt. 0) input_data%qobs = qobs
        if (setup%sparse_storage) then
            do t = 1, nt
                call sparse_matrix_to_vector_r(mesh, prcp(:, :, t), input_data%sparse_prcp(:, t))
                call sparse_matrix_to_vector_r(mesh, pet(:, :, t), input_data%sparse_pet(:, t))
            end do
        else
            input_data%prcp = prcp
            input_data%pet = pet
        end if

        if (icfg(14) .gt. 0 .and. present(descriptor)) input_data%descriptor = descriptor(:, :, 1:icfg(14))
        call ParametersDT_initialise(p, mesh)
        call ParametersDT_initialise(p_b, mesh)
        call ParametersDT_initialise(p_bgd, mesh)
        call ParametersDT_initialise(p_bgd_b, mesh)
        call StatesDT_initialise(s, mesh)
        call StatesDT_initialise(s_b, mesh)
        call StatesDT_initialise(s_bgd, mesh)
        call StatesDT_initialise(s_bgd_b, mesh)
        call OutputDT_initialise(output, setup, mesh)
        call OutputDT_initialise(output_b, setup, mesh)

        !  Optimize_SetupDT fields the Python caller would set (smash/core/simulation/_optimize.py:173-229)
        setup%optimize%denormalize_forward = (icfg(7) .ne. 0)
        setup%optimize%optimize_start_step = icfg(8)
        setup%optimize%njf = njf
        setup%optimize%njr = njr
        deallocate (setup%optimize%jobs_fun, setup%optimize%wjobs_fun)
        deallocate (setup%optimize%jreg_fun, setup%optimize%wjreg_fun)
        allocate (setup%optimize%jobs_fun(njf), setup%optimize%wjobs_fun(njf))
        allocate (setup%optimize%jreg_fun(njr), setup%optimize%wjreg_fun(njr))
        do j = 1, njf
            setup%optimize%jobs_fun(j) = jobs_name(jobs_codes(j))
            setup%optimize%wjobs_fun(j) = wjobs(j)
        end do
        do j = 1, njr
            setup%optimize%jreg_fun(j) = jreg_name(jreg_codes(j))
            setup%optimize%wjreg_fun(j) = wjreg_fun(j)
        end do
        setup%optimize%wjreg = rcfg(3)
        if (ng .gt. 0) setup%optimize%wgauge = wgauge
        setup%optimize%optim_parameters = optim_p
        setup%optimize%optim_states = optim_s
        setup%optimize%lb_parameters = lbp
        setup%optimize%ub_parameters = ubp
        setup%optimize%lb_states = lbs
        setup%optimize%ub_states = ubs

        call set_parameters(mesh, p_bgd, params_bgd)
        call set_states(mesh, s_bgd, states_bgd)

        elapsed = 0._c_double
        do rep = 1, nrep
            call set_parameters(mesh, p, params)
            call set_states(mesh, s, states)
            cost = 0._sp
            cost_b = rcfg(4)
            call system_clock(c0, crate)
            if (icfg(11) .eq. 0) then
                call forward(setup, mesh, input_data, p, p_bgd, s, s_bgd, output, cost)
            else if (icfg(11) .ge. 4 .and. icfg(11) .le. 6) then
                call Hyper_ParametersDT_initialise(hp, setup)
                call Hyper_ParametersDT_initialise(hp_b, setup)
                call Hyper_ParametersDT_initialise(hp_bgd, setup)
                call Hyper_StatesDT_initialise(hs, setup)
                call Hyper_StatesDT_initialise(hs_b, setup)
                call Hyper_StatesDT_initialise(hs_bgd, setup)
                call set_hyper_parameters(setup, hp, hyper_p)
                call set_hyper_parameters(setup, hp_bgd, hyper_p)
                call set_hyper_states(setup, hs, hyper_s)
                call set_hyper_states(setup, hs_bgd, hyper_s)
                if (icfg(11) .eq. 4) then
                    call hyper_forward(setup, mesh, input_data, p, hp, hp_bgd, s, hs, hs_bgd, output, cost)
                else if (icfg(11) .eq. 6) then
                    call ParametersDT_initialise(p_d, mesh)
                    call StatesDT_initialise(s_d, mesh)
                    call OutputDT_initialise(output_d, setup, mesh)
                    call set_hyper_parameters(setup, hp_b, hyper_p_b)
                    call set_hyper_states(setup, hs_b, hyper_s_b)
                    cost_d = 0._sp
                    call hyper_forward_d(setup, mesh, input_data, p, p_d, hp, hp_b, hp_bgd, s, s_d, hs, hs_b, hs_bgd, &
                    & output, output_d, cost, cost_d)
                    cost_d_out = cost_d
                    if (ng .gt. 0) qsim_d = output_d%qsim
                else
                    call hyper_forward_b(setup, mesh, input_data, p, p_b, hp, hp_b, hp_bgd, s, s_b, hs, hs_b, hs_bgd, &
                    & output, output_b, cost, cost_b)
                    call get_hyper_parameters(setup, hp_b, hyper_p_b)
                    call get_hyper_states(setup, hs_b, hyper_s_b)
                end if
            else if (icfg(11) .eq. 3) then
                !  tangent model: p_b / s_b objects double as the (scratch) background tangents
                call ParametersDT_initialise(p_d, mesh)
                call StatesDT_initialise(s_d, mesh)
                call OutputDT_initialise(output_d, setup, mesh)
                call set_parameters(mesh, p_d, params_d)
                call set_states(mesh, s_d, states_d)
                call set_parameters(mesh, p_bgd_b, params_bgd_d)
                call set_states(mesh, s_bgd_b, states_bgd_d)
                cost_d = 0._sp
                call forward_d(setup, mesh, input_data, p, p_d, p_bgd, p_bgd_b, &
                & s, s_d, s_bgd, s_bgd_b, output, output_d, cost, cost_d)
                cost_d_out = cost_d
                if (ng .gt. 0) qsim_d = output_d%qsim
            else if (icfg(11) .eq. 7) then
                !  mw_optimize::optimize_sbs (mw_optimize.f90:53-294): the uniform step-by-step calibration, forward sweeps only
                setup%optimize%maxiter = icfg(13)
                setup%optimize%verbose = .false.
                call optimize_sbs(setup, mesh, input_data, p, s, output)
                cost = output%cost
            else if (icfg(11) .eq. 2) then
                setup%optimize%denormalize_forward = .false.
                setup%optimize%maxiter = icfg(13)
                setup%optimize%verbose = .false.
                call optimize_lbfgsb(setup, mesh, input_data, p, s, output)
                cost = output%cost
            else
                call forward_b(setup, mesh, input_data, p, p_b, p_bgd, p_bgd_b, &
                & s, s_b, s_bgd, s_bgd_b, output, output_b, cost, cost_b)
            end if
            call system_clock(c1)
            elapsed = elapsed + real(c1 - c0, c_double)/real(crate, c_double)
        end do
        elapsed = elapsed/real(nrep, c_double)

        if (ng .gt. 0) qsim = output%qsim
        costs(1) = cost
        costs(2) = output%cost_jobs
        costs(3) = output%cost_jreg
        call get_states(mesh, output%fstates, fstates)
        call get_parameters(mesh, p, params_out)
        call get_states(mesh, s, states_out)
        if (icfg(11) .eq. 1) then
            call get_parameters(mesh, p_b, params_b)
            call get_states(mesh, s_b, states_b)
        end if

    end subroutine ref_core

    subroutine ref_run(icfg, rcfg, flwdir, flwacc, path, active_cell, gauge_pos, area, &
    & prcp, pet, qobs, params, params_bgd, states, states_bgd, &
    & wgauge, jobs_codes, wjobs, jreg_codes, wjreg_fun, optim_p, optim_s, lbp, ubp, lbs, ubs, &
    & qsim, costs, fstates, params_out, states_out, params_b, states_b, elapsed) bind(C, name="ref_run")
        integer(c_int), intent(in) :: icfg(16)
        real(c_float), intent(in) :: rcfg(4)
        integer(c_int), intent(in) :: flwdir(icfg(2), icfg(3)), flwacc(icfg(2), icfg(3))
        integer(c_int), intent(in) :: path(2, icfg(2)*icfg(3)), active_cell(icfg(2), icfg(3))
        integer(c_int), intent(in) :: gauge_pos(icfg(5), 2)
        real(c_float), intent(in) :: area(icfg(5))
        real(c_float), intent(in) :: prcp(icfg(2), icfg(3), icfg(4)), pet(icfg(2), icfg(3), icfg(4))
        real(c_float), intent(in) :: qobs(icfg(5), icfg(4))
        real(c_float), intent(in) :: params(icfg(2), icfg(3), GNP), params_bgd(icfg(2), icfg(3), GNP)
        real(c_float), intent(in) :: states(icfg(2), icfg(3), GNS), states_bgd(icfg(2), icfg(3), GNS)
        real(c_float), intent(in) :: wgauge(icfg(5))
        integer(c_int), intent(in) :: jobs_codes(*)
        real(c_float), intent(in) :: wjobs(*)
        integer(c_int), intent(in) :: jreg_codes(*)
        real(c_float), intent(in) :: wjreg_fun(*)
        integer(c_int), intent(in) :: optim_p(GNP), optim_s(GNS)
        real(c_float), intent(in) :: lbp(GNP), ubp(GNP), lbs(GNS), ubs(GNS)
        real(c_float), intent(inout) :: qsim(icfg(5), icfg(4))
        real(c_float), intent(inout) :: costs(3)
        real(c_float), intent(inout) :: fstates(icfg(2), icfg(3), GNS)
        real(c_float), intent(inout) :: params_out(icfg(2), icfg(3), GNP), states_out(icfg(2), icfg(3), GNS)
        real(c_float), intent(inout) :: params_b(icfg(2), icfg(3), GNP), states_b(icfg(2), icfg(3), GNS)
        real(c_double), intent(inout) :: elapsed
        call ref_core(icfg, rcfg, flwdir, flwacc, path, active_cell, gauge_pos, area, &
        & prcp, pet, qobs, params, params_bgd, states, states_bgd, &
        & wgauge, jobs_codes, wjobs, jreg_codes, wjreg_fun, optim_p, optim_s, lbp, ubp, lbs, ubs, &
        & qsim, costs, fstates, params_out, states_out, params_b, states_b, elapsed)
    end subroutine ref_run

    !  icfg(11) must be 3: mw_forward::forward_d with the tangent direction (params_d, states_d) and the tangents
    !  of the background fields; returns output_d%qsim and cost_d beside the primal outputs
    subroutine ref_run_d(icfg, rcfg, flwdir, flwacc, path, active_cell, gauge_pos, area, &
    & prcp, pet, qobs, params, params_bgd, states, states_bgd, &
    & wgauge, jobs_codes, wjobs, jreg_codes, wjreg_fun, optim_p, optim_s, lbp, ubp, lbs, ubs, &
    & qsim, costs, fstates, params_out, states_out, params_b, states_b, elapsed, &
    & params_d, states_d, params_bgd_d, states_bgd_d, qsim_d, cost_d) bind(C, name="ref_run_d")
        integer(c_int), intent(in) :: icfg(16)
        real(c_float), intent(in) :: rcfg(4)
        integer(c_int), intent(in) :: flwdir(icfg(2), icfg(3)), flwacc(icfg(2), icfg(3))
        integer(c_int), intent(in) :: path(2, icfg(2)*icfg(3)), active_cell(icfg(2), icfg(3))
        integer(c_int), intent(in) :: gauge_pos(icfg(5), 2)
        real(c_float), intent(in) :: area(icfg(5))
        real(c_float), intent(in) :: prcp(icfg(2), icfg(3), icfg(4)), pet(icfg(2), icfg(3), icfg(4))
        real(c_float), intent(in) :: qobs(icfg(5), icfg(4))
        real(c_float), intent(in) :: params(icfg(2), icfg(3), GNP), params_bgd(icfg(2), icfg(3), GNP)
        real(c_float), intent(in) :: states(icfg(2), icfg(3), GNS), states_bgd(icfg(2), icfg(3), GNS)
        real(c_float), intent(in) :: wgauge(icfg(5))
        integer(c_int), intent(in) :: jobs_codes(*)
        real(c_float), intent(in) :: wjobs(*)
        integer(c_int), intent(in) :: jreg_codes(*)
        real(c_float), intent(in) :: wjreg_fun(*)
        integer(c_int), intent(in) :: optim_p(GNP), optim_s(GNS)
        real(c_float), intent(in) :: lbp(GNP), ubp(GNP), lbs(GNS), ubs(GNS)
        real(c_float), intent(inout) :: qsim(icfg(5), icfg(4))
        real(c_float), intent(inout) :: costs(3)
        real(c_float), intent(inout) :: fstates(icfg(2), icfg(3), GNS)
        real(c_float), intent(inout) :: params_out(icfg(2), icfg(3), GNP), states_out(icfg(2), icfg(3), GNS)
        real(c_float), intent(inout) :: params_b(icfg(2), icfg(3), GNP), states_b(icfg(2), icfg(3), GNS)
        real(c_double), intent(inout) :: elapsed
        real(c_float), intent(in) :: params_d(icfg(2), icfg(3), GNP), states_d(icfg(2), icfg(3), GNS)
        real(c_float), intent(in) :: params_bgd_d(icfg(2), icfg(3), GNP), states_bgd_d(icfg(2), icfg(3), GNS)
        real(c_float), intent(inout) :: qsim_d(icfg(5), icfg(4))
        real(c_float), intent(inout) :: cost_d
        call ref_core(icfg, rcfg, flwdir, flwacc, path, active_cell, gauge_pos, area, &
        & prcp, pet, qobs, params, params_bgd, states, states_bgd, &
        & wgauge, jobs_codes, wjobs, jreg_codes, wjreg_fun, optim_p, optim_s, lbp, ubp, lbs, ubs, &
        & qsim, costs, fstates, params_out, states_out, params_b, states_b, elapsed, &
        & params_d, states_d, params_bgd_d, states_bgd_d, qsim_d, cost_d)
    end subroutine ref_run_d

    !  icfg(11) = 4 / 5: mw_forward::hyper_forward / hyper_forward_b (descriptor -> parameter mappings)
    subroutine ref_run_hyper(icfg, rcfg, flwdir, flwacc, path, active_cell, gauge_pos, area, &
    & prcp, pet, qobs, params, params_bgd, states, states_bgd, &
    & wgauge, jobs_codes, wjobs, jreg_codes, wjreg_fun, optim_p, optim_s, lbp, ubp, lbs, ubs, &
    & qsim, costs, fstates, params_out, states_out, params_b, states_b, elapsed, &
    & descriptor, hyper_p, hyper_s, hyper_p_b, hyper_s_b, qsim_d, cost_d_out) bind(C, name="ref_run_hyper")
        integer(c_int), intent(in) :: icfg(16)
        real(c_float), intent(in) :: rcfg(4)
        integer(c_int), intent(in) :: flwdir(icfg(2), icfg(3)), flwacc(icfg(2), icfg(3))
        integer(c_int), intent(in) :: path(2, icfg(2)*icfg(3)), active_cell(icfg(2), icfg(3))
        integer(c_int), intent(in) :: gauge_pos(icfg(5), 2)
        real(c_float), intent(in) :: area(icfg(5))
        real(c_float), intent(in) :: prcp(icfg(2), icfg(3), icfg(4)), pet(icfg(2), icfg(3), icfg(4))
        real(c_float), intent(in) :: qobs(icfg(5), icfg(4))
        real(c_float), intent(in) :: params(icfg(2), icfg(3), GNP), params_bgd(icfg(2), icfg(3), GNP)
        real(c_float), intent(in) :: states(icfg(2), icfg(3), GNS), states_bgd(icfg(2), icfg(3), GNS)
        real(c_float), intent(in) :: wgauge(icfg(5))
        integer(c_int), intent(in) :: jobs_codes(*)
        real(c_float), intent(in) :: wjobs(*)
        integer(c_int), intent(in) :: jreg_codes(*)
        real(c_float), intent(in) :: wjreg_fun(*)
        integer(c_int), intent(in) :: optim_p(GNP), optim_s(GNS)
        real(c_float), intent(in) :: lbp(GNP), ubp(GNP), lbs(GNS), ubs(GNS)
        real(c_float), intent(inout) :: qsim(icfg(5), icfg(4))
        real(c_float), intent(inout) :: costs(3)
        real(c_float), intent(inout) :: fstates(icfg(2), icfg(3), GNS)
        real(c_float), intent(inout) :: params_out(icfg(2), icfg(3), GNP), states_out(icfg(2), icfg(3), GNS)
        real(c_float), intent(inout) :: params_b(icfg(2), icfg(3), GNP), states_b(icfg(2), icfg(3), GNS)
        real(c_double), intent(inout) :: elapsed
        real(c_float), intent(in) :: descriptor(icfg(2), icfg(3), *)
        real(c_float), intent(in) :: hyper_p(1 + icfg(15)*icfg(14), 1, GNP), hyper_s(1 + icfg(15)*icfg(14), 1, GNS)
        real(c_float), intent(inout) :: hyper_p_b(1 + icfg(15)*icfg(14), 1, GNP), hyper_s_b(1 + icfg(15)*icfg(14), 1, GNS)
        real(c_float), intent(inout) :: qsim_d(icfg(5), icfg(4))
        real(c_float), intent(inout) :: cost_d_out
        call ref_core(icfg, rcfg, flwdir, flwacc, path, active_cell, gauge_pos, area, &
        & prcp, pet, qobs, params, params_bgd, states, states_bgd, &
        & wgauge, jobs_codes, wjobs, jreg_codes, wjreg_fun, optim_p, optim_s, lbp, ubp, lbs, ubs, &
        & qsim, costs, fstates, params_out, states_out, params_b, states_b, elapsed, &
        & qsim_d=qsim_d, cost_d_out=cost_d_out, &
        & descriptor=descriptor, hyper_p=hyper_p, hyper_s=hyper_s, hyper_p_b=hyper_p_b, hyper_s_b=hyper_s_b)
    end subroutine ref_run_hyper

end module ref_capi
