﻿!mod$ v1 sum:b4ff4a9a3b3e22fd
!need$ 0bde2ac47243ead2 i iso_c_binding
!need$ b74288d896965ed5 n mw_sparse_storage
!need$ 5646c8d7b79c7ea8 n mw_optimize
!need$ 86b12428149ac79a n md_constant
!need$ 82a26416841665dd n mwd_setup
!need$ b7e498e07543ba78 n mwd_mesh
!need$ 76c1396aa4cc0721 n mwd_input_data
!need$ eda5fd194b829f52 n mwd_parameters
!need$ c5f5068eb58aec21 n mwd_states
!need$ 44a770df04028c8f n mwd_output
!need$ 669a078b745dce26 n mw_forward
!need$ 4d57810507808050 n mwd_parameters_manipulation
!need$ ba196de66a1b48a6 n mwd_states_manipulation
module ref_capi
use mw_forward,only:forward
use mw_forward,only:forward_b
use mw_forward,only:forward_d
use mw_forward,only:hyper_forward
use mw_forward,only:hyper_forward_b
use mw_optimize,only:optimize_lbfgsb
use,intrinsic::iso_c_binding,only:c_associated
use,intrinsic::iso_c_binding,only:c_funloc
use,intrinsic::iso_c_binding,only:c_funptr
use,intrinsic::iso_c_binding,only:c_f_pointer
use,intrinsic::iso_c_binding,only:c_loc
use,intrinsic::iso_c_binding,only:c_null_funptr
use,intrinsic::iso_c_binding,only:c_null_ptr
use,intrinsic::iso_c_binding,only:c_ptr
use,intrinsic::iso_c_binding,only:c_sizeof
use,intrinsic::iso_c_binding,only:operator(==)
use,intrinsic::iso_c_binding,only:operator(/=)
use,intrinsic::iso_c_binding,only:c_int8_t
use,intrinsic::iso_c_binding,only:c_int16_t
use,intrinsic::iso_c_binding,only:c_int32_t
use,intrinsic::iso_c_binding,only:c_int64_t
use,intrinsic::iso_c_binding,only:c_int128_t
use,intrinsic::iso_c_binding,only:c_int
use,intrinsic::iso_c_binding,only:c_short
use,intrinsic::iso_c_binding,only:c_long
use,intrinsic::iso_c_binding,only:c_long_long
use,intrinsic::iso_c_binding,only:c_signed_char
use,intrinsic::iso_c_binding,only:c_size_t
use,intrinsic::iso_c_binding,only:c_intmax_t
use,intrinsic::iso_c_binding,only:c_intptr_t
use,intrinsic::iso_c_binding,only:c_ptrdiff_t
use,intrinsic::iso_c_binding,only:c_int_least8_t
use,intrinsic::iso_c_binding,only:c_int_fast8_t
use,intrinsic::iso_c_binding,only:c_int_least16_t
use,intrinsic::iso_c_binding,only:c_int_fast16_t
use,intrinsic::iso_c_binding,only:c_int_least32_t
use,intrinsic::iso_c_binding,only:c_int_fast32_t
use,intrinsic::iso_c_binding,only:c_int_least64_t
use,intrinsic::iso_c_binding,only:c_int_fast64_t
use,intrinsic::iso_c_binding,only:c_int_least128_t
use,intrinsic::iso_c_binding,only:c_int_fast128_t
use,intrinsic::iso_c_binding,only:c_float
use,intrinsic::iso_c_binding,only:c_double
use,intrinsic::iso_c_binding,only:c_long_double
use,intrinsic::iso_c_binding,only:c_float_complex
use,intrinsic::iso_c_binding,only:c_double_complex
use,intrinsic::iso_c_binding,only:c_long_double_complex
use,intrinsic::iso_c_binding,only:c_bool
use,intrinsic::iso_c_binding,only:c_char
use,intrinsic::iso_c_binding,only:c_null_char
use,intrinsic::iso_c_binding,only:c_alert
use,intrinsic::iso_c_binding,only:c_backspace
use,intrinsic::iso_c_binding,only:c_form_feed
use,intrinsic::iso_c_binding,only:c_new_line
use,intrinsic::iso_c_binding,only:c_carriage_return
use,intrinsic::iso_c_binding,only:c_horizontal_tab
use,intrinsic::iso_c_binding,only:c_vertical_tab
use,intrinsic::iso_c_binding,only:c_float128
use,intrinsic::iso_c_binding,only:c_float128_complex
use,intrinsic::iso_c_binding,only:c_uint8_t
use,intrinsic::iso_c_binding,only:c_uint16_t
use,intrinsic::iso_c_binding,only:c_uint32_t
use,intrinsic::iso_c_binding,only:c_uint64_t
use,intrinsic::iso_c_binding,only:c_uint128_t
use,intrinsic::iso_c_binding,only:c_unsigned_char
use,intrinsic::iso_c_binding,only:c_unsigned_short
use,intrinsic::iso_c_binding,only:c_unsigned
use,intrinsic::iso_c_binding,only:c_unsigned_long
use,intrinsic::iso_c_binding,only:c_unsigned_long_long
use,intrinsic::iso_c_binding,only:c_uintmax_t
use,intrinsic::iso_c_binding,only:c_uint_fast8_t
use,intrinsic::iso_c_binding,only:c_uint_fast16_t
use,intrinsic::iso_c_binding,only:c_uint_fast32_t
use,intrinsic::iso_c_binding,only:c_uint_fast64_t
use,intrinsic::iso_c_binding,only:c_uint_fast128_t
use,intrinsic::iso_c_binding,only:c_uint_least8_t
use,intrinsic::iso_c_binding,only:c_uint_least16_t
use,intrinsic::iso_c_binding,only:c_uint_least32_t
use,intrinsic::iso_c_binding,only:c_uint_least64_t
use,intrinsic::iso_c_binding,only:c_uint_least128_t
use,intrinsic::iso_c_binding,only:c_f_procpointer
use md_constant,only:sp
use md_constant,only:dp
use md_constant,only:lchar
use md_constant,only:gnp
use md_constant,only:gns
use md_constant,only:gparameters_name
use md_constant,only:gstates_name
use md_constant,only:glb_parameters
use md_constant,only:gub_parameters
use md_constant,only:glb_states
use md_constant,only:gub_states
use mwd_setup,only:optimize_setupdt
use mwd_setup,only:setupdt
use mwd_setup,only:optimize_setupdt_initialise
use mwd_setup,only:setupdt_initialise
use mwd_mesh,only:meshdt
use mwd_mesh,only:meshdt_initialise
use mwd_input_data,only:input_datadt
use mwd_input_data,only:input_datadt_initialise
use mwd_parameters,only:parametersdt
use mwd_parameters,only:hyper_parametersdt
use mwd_parameters,only:parametersdt_initialise
use mwd_parameters,only:hyper_parametersdt_initialise
use mwd_states,only:statesdt
use mwd_states,only:hyper_statesdt
use mwd_states,only:statesdt_initialise
use mwd_states,only:hyper_statesdt_initialise
use mwd_output,only:outputdt
use mwd_output,only:outputdt_initialise
use mwd_parameters_manipulation,only:set_parameters
use mwd_parameters_manipulation,only:set_hyper_parameters
use mwd_parameters_manipulation,only:get_parameters
use mwd_parameters_manipulation,only:set3d_parameters
use mwd_parameters_manipulation,only:set1d_parameters
use mwd_parameters_manipulation,only:set0d_parameters
use mwd_parameters_manipulation,only:normalize_parameters
use mwd_parameters_manipulation,only:denormalize_parameters
use mwd_parameters_manipulation,only:get_hyper_parameters
use mwd_parameters_manipulation,only:set3d_hyper_parameters
use mwd_parameters_manipulation,only:set1d_hyper_parameters
use mwd_parameters_manipulation,only:set0d_hyper_parameters
use mwd_parameters_manipulation,only:hyper_parameters_to_parameters
use mwd_states_manipulation,only:set_states
use mwd_states_manipulation,only:set_hyper_states
use mwd_states_manipulation,only:get_states
use mwd_states_manipulation,only:set3d_states
use mwd_states_manipulation,only:set1d_states
use mwd_states_manipulation,only:set0d_states
use mwd_states_manipulation,only:normalize_states
use mwd_states_manipulation,only:denormalize_states
use mwd_states_manipulation,only:get_hyper_states
use mwd_states_manipulation,only:set3d_hyper_states
use mwd_states_manipulation,only:set1d_hyper_states
use mwd_states_manipulation,only:set0d_hyper_states
use mwd_states_manipulation,only:hyper_states_to_states
use mw_sparse_storage,only:compute_rowcol_to_ind_sparse
use mw_sparse_storage,only:sparse_matrix_to_vector_r
use mw_sparse_storage,only:sparse_matrix_to_vector_i
use mw_sparse_storage,only:sparse_vector_to_matrix_r
use mw_sparse_storage,only:sparse_vector_to_matrix_i
contains
function jobs_name(code) result(s)
integer(4),intent(in)::code
character(20_4,1)::s
end
function jreg_name(code) result(s)
integer(4),intent(in)::code
character(20_4,1)::s
end
subroutine ref_core(icfg,rcfg,flwdir,flwacc,path,active_cell,gauge_pos,area,prcp,pet,qobs,params,params_bgd,states,states_bgd,wgauge,jobs_codes,wjobs,jreg_codes,wjreg_fun,optim_p,optim_s,lbp,ubp,lbs,ubs,qsim,costs,fstates,params_out,states_out,params_b,states_b,elapsed,params_d,states_d,params_bgd_d,states_bgd_d,qsim_d,cost_d_out,descriptor,hyper_p,hyper_s,hyper_p_b,hyper_s_b)
integer(4),intent(in)::icfg(1_8:16_8)
real(4),intent(in)::rcfg(1_8:4_8)
integer(4),intent(in)::flwdir(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8))
integer(4),intent(in)::flwacc(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8))
integer(4),intent(in)::path(1_8:2_8,1_8:int(icfg(2_8)*icfg(3_8),kind=8))
integer(4),intent(in)::active_cell(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8))
integer(4),intent(in)::gauge_pos(1_8:int(icfg(5_8),kind=8),1_8:2_8)
real(4),intent(in)::area(1_8:int(icfg(5_8),kind=8))
real(4),intent(in)::prcp(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:int(icfg(4_8),kind=8))
real(4),intent(in)::pet(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:int(icfg(4_8),kind=8))
real(4),intent(in)::qobs(1_8:int(icfg(5_8),kind=8),1_8:int(icfg(4_8),kind=8))
real(4),intent(in)::params(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:16_8)
real(4),intent(in)::params_bgd(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:16_8)
real(4),intent(in)::states(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:8_8)
real(4),intent(in)::states_bgd(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:8_8)
real(4),intent(in)::wgauge(1_8:int(icfg(5_8),kind=8))
integer(4),intent(in)::jobs_codes(1_8:*)
real(4),intent(in)::wjobs(1_8:*)
integer(4),intent(in)::jreg_codes(1_8:*)
real(4),intent(in)::wjreg_fun(1_8:*)
integer(4),intent(in)::optim_p(1_8:16_8)
integer(4),intent(in)::optim_s(1_8:8_8)
real(4),intent(in)::lbp(1_8:16_8)
real(4),intent(in)::ubp(1_8:16_8)
real(4),intent(in)::lbs(1_8:8_8)
real(4),intent(in)::ubs(1_8:8_8)
real(4),intent(inout)::qsim(1_8:int(icfg(5_8),kind=8),1_8:int(icfg(4_8),kind=8))
real(4),intent(inout)::costs(1_8:3_8)
real(4),intent(inout)::fstates(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:8_8)
real(4),intent(inout)::params_out(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:16_8)
real(4),intent(inout)::states_out(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:8_8)
real(4),intent(inout)::params_b(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:16_8)
real(4),intent(inout)::states_b(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:8_8)
real(8),intent(inout)::elapsed
real(4),intent(in),optional::params_d(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:16_8)
real(4),intent(in),optional::states_d(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:8_8)
real(4),intent(in),optional::params_bgd_d(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:16_8)
real(4),intent(in),optional::states_bgd_d(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:8_8)
real(4),intent(inout),optional::qsim_d(1_8:int(icfg(5_8),kind=8),1_8:int(icfg(4_8),kind=8))
real(4),intent(inout),optional::cost_d_out
real(4),intent(in),optional::descriptor(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:*)
real(4),intent(in),optional::hyper_p(1_8:int(1_4+icfg(15_8)*icfg(14_8),kind=8),1_8:1_8,1_8:16_8)
real(4),intent(in),optional::hyper_s(1_8:int(1_4+icfg(15_8)*icfg(14_8),kind=8),1_8:1_8,1_8:8_8)
real(4),intent(inout),optional::hyper_p_b(1_8:int(1_4+icfg(15_8)*icfg(14_8),kind=8),1_8:1_8,1_8:16_8)
real(4),intent(inout),optional::hyper_s_b(1_8:int(1_4+icfg(15_8)*icfg(14_8),kind=8),1_8:1_8,1_8:8_8)
end
subroutine ref_run(icfg,rcfg,flwdir,flwacc,path,active_cell,gauge_pos,area,prcp,pet,qobs,params,params_bgd,states,states_bgd,wgauge,jobs_codes,wjobs,jreg_codes,wjreg_fun,optim_p,optim_s,lbp,ubp,lbs,ubs,qsim,costs,fstates,params_out,states_out,params_b,states_b,elapsed) bind(c,name="ref_run")
integer(4),intent(in)::icfg(1_8:16_8)
real(4),intent(in)::rcfg(1_8:4_8)
integer(4),intent(in)::flwdir(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8))
integer(4),intent(in)::flwacc(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8))
integer(4),intent(in)::path(1_8:2_8,1_8:int(icfg(2_8)*icfg(3_8),kind=8))
integer(4),intent(in)::active_cell(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8))
integer(4),intent(in)::gauge_pos(1_8:int(icfg(5_8),kind=8),1_8:2_8)
real(4),intent(in)::area(1_8:int(icfg(5_8),kind=8))
real(4),intent(in)::prcp(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:int(icfg(4_8),kind=8))
real(4),intent(in)::pet(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:int(icfg(4_8),kind=8))
real(4),intent(in)::qobs(1_8:int(icfg(5_8),kind=8),1_8:int(icfg(4_8),kind=8))
real(4),intent(in)::params(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:16_8)
real(4),intent(in)::params_bgd(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:16_8)
real(4),intent(in)::states(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:8_8)
real(4),intent(in)::states_bgd(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:8_8)
real(4),intent(in)::wgauge(1_8:int(icfg(5_8),kind=8))
integer(4),intent(in)::jobs_codes(1_8:*)
real(4),intent(in)::wjobs(1_8:*)
integer(4),intent(in)::jreg_codes(1_8:*)
real(4),intent(in)::wjreg_fun(1_8:*)
integer(4),intent(in)::optim_p(1_8:16_8)
integer(4),intent(in)::optim_s(1_8:8_8)
real(4),intent(in)::lbp(1_8:16_8)
real(4),intent(in)::ubp(1_8:16_8)
real(4),intent(in)::lbs(1_8:8_8)
real(4),intent(in)::ubs(1_8:8_8)
real(4),intent(inout)::qsim(1_8:int(icfg(5_8),kind=8),1_8:int(icfg(4_8),kind=8))
real(4),intent(inout)::costs(1_8:3_8)
real(4),intent(inout)::fstates(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:8_8)
real(4),intent(inout)::params_out(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:16_8)
real(4),intent(inout)::states_out(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:8_8)
real(4),intent(inout)::params_b(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:16_8)
real(4),intent(inout)::states_b(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:8_8)
real(8),intent(inout)::elapsed
end
subroutine ref_run_d(icfg,rcfg,flwdir,flwacc,path,active_cell,gauge_pos,area,prcp,pet,qobs,params,params_bgd,states,states_bgd,wgauge,jobs_codes,wjobs,jreg_codes,wjreg_fun,optim_p,optim_s,lbp,ubp,lbs,ubs,qsim,costs,fstates,params_out,states_out,params_b,states_b,elapsed,params_d,states_d,params_bgd_d,states_bgd_d,qsim_d,cost_d) bind(c,name="ref_run_d")
integer(4),intent(in)::icfg(1_8:16_8)
real(4),intent(in)::rcfg(1_8:4_8)
integer(4),intent(in)::flwdir(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8))
integer(4),intent(in)::flwacc(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8))
integer(4),intent(in)::path(1_8:2_8,1_8:int(icfg(2_8)*icfg(3_8),kind=8))
integer(4),intent(in)::active_cell(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8))
integer(4),intent(in)::gauge_pos(1_8:int(icfg(5_8),kind=8),1_8:2_8)
real(4),intent(in)::area(1_8:int(icfg(5_8),kind=8))
real(4),intent(in)::prcp(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:int(icfg(4_8),kind=8))
real(4),intent(in)::pet(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:int(icfg(4_8),kind=8))
real(4),intent(in)::qobs(1_8:int(icfg(5_8),kind=8),1_8:int(icfg(4_8),kind=8))
real(4),intent(in)::params(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:16_8)
real(4),intent(in)::params_bgd(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:16_8)
real(4),intent(in)::states(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:8_8)
real(4),intent(in)::states_bgd(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:8_8)
real(4),intent(in)::wgauge(1_8:int(icfg(5_8),kind=8))
integer(4),intent(in)::jobs_codes(1_8:*)
real(4),intent(in)::wjobs(1_8:*)
integer(4),intent(in)::jreg_codes(1_8:*)
real(4),intent(in)::wjreg_fun(1_8:*)
integer(4),intent(in)::optim_p(1_8:16_8)
integer(4),intent(in)::optim_s(1_8:8_8)
real(4),intent(in)::lbp(1_8:16_8)
real(4),intent(in)::ubp(1_8:16_8)
real(4),intent(in)::lbs(1_8:8_8)
real(4),intent(in)::ubs(1_8:8_8)
real(4),intent(inout)::qsim(1_8:int(icfg(5_8),kind=8),1_8:int(icfg(4_8),kind=8))
real(4),intent(inout)::costs(1_8:3_8)
real(4),intent(inout)::fstates(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:8_8)
real(4),intent(inout)::params_out(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:16_8)
real(4),intent(inout)::states_out(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:8_8)
real(4),intent(inout)::params_b(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:16_8)
real(4),intent(inout)::states_b(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:8_8)
real(8),intent(inout)::elapsed
real(4),intent(in)::params_d(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:16_8)
real(4),intent(in)::states_d(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:8_8)
real(4),intent(in)::params_bgd_d(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:16_8)
real(4),intent(in)::states_bgd_d(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:8_8)
real(4),intent(inout)::qsim_d(1_8:int(icfg(5_8),kind=8),1_8:int(icfg(4_8),kind=8))
real(4),intent(inout)::cost_d
end
subroutine ref_run_hyper(icfg,rcfg,flwdir,flwacc,path,active_cell,gauge_pos,area,prcp,pet,qobs,params,params_bgd,states,states_bgd,wgauge,jobs_codes,wjobs,jreg_codes,wjreg_fun,optim_p,optim_s,lbp,ubp,lbs,ubs,qsim,costs,fstates,params_out,states_out,params_b,states_b,elapsed,descriptor,hyper_p,hyper_s,hyper_p_b,hyper_s_b) bind(c,name="ref_run_hyper")
integer(4),intent(in)::icfg(1_8:16_8)
real(4),intent(in)::rcfg(1_8:4_8)
integer(4),intent(in)::flwdir(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8))
integer(4),intent(in)::flwacc(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8))
integer(4),intent(in)::path(1_8:2_8,1_8:int(icfg(2_8)*icfg(3_8),kind=8))
integer(4),intent(in)::active_cell(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8))
integer(4),intent(in)::gauge_pos(1_8:int(icfg(5_8),kind=8),1_8:2_8)
real(4),intent(in)::area(1_8:int(icfg(5_8),kind=8))
real(4),intent(in)::prcp(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:int(icfg(4_8),kind=8))
real(4),intent(in)::pet(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:int(icfg(4_8),kind=8))
real(4),intent(in)::qobs(1_8:int(icfg(5_8),kind=8),1_8:int(icfg(4_8),kind=8))
real(4),intent(in)::params(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:16_8)
real(4),intent(in)::params_bgd(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:16_8)
real(4),intent(in)::states(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:8_8)
real(4),intent(in)::states_bgd(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:8_8)
real(4),intent(in)::wgauge(1_8:int(icfg(5_8),kind=8))
integer(4),intent(in)::jobs_codes(1_8:*)
real(4),intent(in)::wjobs(1_8:*)
integer(4),intent(in)::jreg_codes(1_8:*)
real(4),intent(in)::wjreg_fun(1_8:*)
integer(4),intent(in)::optim_p(1_8:16_8)
integer(4),intent(in)::optim_s(1_8:8_8)
real(4),intent(in)::lbp(1_8:16_8)
real(4),intent(in)::ubp(1_8:16_8)
real(4),intent(in)::lbs(1_8:8_8)
real(4),intent(in)::ubs(1_8:8_8)
real(4),intent(inout)::qsim(1_8:int(icfg(5_8),kind=8),1_8:int(icfg(4_8),kind=8))
real(4),intent(inout)::costs(1_8:3_8)
real(4),intent(inout)::fstates(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:8_8)
real(4),intent(inout)::params_out(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:16_8)
real(4),intent(inout)::states_out(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:8_8)
real(4),intent(inout)::params_b(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:16_8)
real(4),intent(inout)::states_b(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:8_8)
real(8),intent(inout)::elapsed
real(4),intent(in)::descriptor(1_8:int(icfg(2_8),kind=8),1_8:int(icfg(3_8),kind=8),1_8:*)
real(4),intent(in)::hyper_p(1_8:int(1_4+icfg(15_8)*icfg(14_8),kind=8),1_8:1_8,1_8:16_8)
real(4),intent(in)::hyper_s(1_8:int(1_4+icfg(15_8)*icfg(14_8),kind=8),1_8:1_8,1_8:8_8)
real(4),intent(inout)::hyper_p_b(1_8:int(1_4+icfg(15_8)*icfg(14_8),kind=8),1_8:1_8,1_8:16_8)
real(4),intent(inout)::hyper_s_b(1_8:int(1_4+icfg(15_8)*icfg(14_8),kind=8),1_8:1_8,1_8:8_8)
end
end
