﻿!mod$ v1 sum:43aa38b7ce13498b
!need$ 86b12428149ac79a n md_constant
!need$ 82a26416841665dd n mwd_setup
!need$ b7e498e07543ba78 n mwd_mesh
!need$ 76c1396aa4cc0721 n mwd_input_data
!need$ b688d5c5e1bddd84 n mwd_parameters_diff
!need$ 23b6a22028cfb5af n mwd_states_diff
!need$ 3d7fc12a1465e543 n mwd_output_diff
!need$ 31b450e492b4e7b2 n mwd_parameters_manipulation_diff
!need$ 510fd0f0196ad54c n mwd_states_manipulation_diff
module mwd_cost_diff
use mwd_states_diff,only:statesdt
use mwd_states_diff,only:hyper_statesdt
use mwd_states_diff,only:statesdt_initialise
use mwd_states_diff,only:hyper_statesdt_initialise
use mwd_output_diff,only:outputdt
use mwd_output_diff,only:outputdt_diff
use mwd_output_diff,only:outputdt_initialise
use mwd_parameters_diff,only:parametersdt
use mwd_parameters_diff,only:hyper_parametersdt
use mwd_parameters_diff,only:parametersdt_initialise
use mwd_parameters_diff,only:hyper_parametersdt_initialise
use mwd_parameters_manipulation_diff,only:set_parameters
use mwd_parameters_manipulation_diff,only:set_parameters_d
use mwd_parameters_manipulation_diff,only:set_parameters_b
use mwd_parameters_manipulation_diff,only:set_hyper_parameters
use mwd_parameters_manipulation_diff,only:get_parameters_d
use mwd_parameters_manipulation_diff,only:get_parameters_b
use mwd_parameters_manipulation_diff,only:get_parameters
use mwd_parameters_manipulation_diff,only:set3d_parameters_d
use mwd_parameters_manipulation_diff,only:set3d_parameters_b
use mwd_parameters_manipulation_diff,only:set3d_parameters
use mwd_parameters_manipulation_diff,only:set1d_parameters
use mwd_parameters_manipulation_diff,only:set0d_parameters
use mwd_parameters_manipulation_diff,only:normalize_parameters_d
use mwd_parameters_manipulation_diff,only:normalize_parameters_b
use mwd_parameters_manipulation_diff,only:normalize_parameters
use mwd_parameters_manipulation_diff,only:denormalize_parameters_d
use mwd_parameters_manipulation_diff,only:denormalize_parameters_b
use mwd_parameters_manipulation_diff,only:denormalize_parameters
use mwd_parameters_manipulation_diff,only:get_hyper_parameters_d
use mwd_parameters_manipulation_diff,only:get_hyper_parameters_b
use mwd_parameters_manipulation_diff,only:get_hyper_parameters
use mwd_parameters_manipulation_diff,only:set3d_hyper_parameters
use mwd_parameters_manipulation_diff,only:set1d_hyper_parameters
use mwd_parameters_manipulation_diff,only:set0d_hyper_parameters
use mwd_parameters_manipulation_diff,only:hyper_parameters_to_parameters_d
use mwd_parameters_manipulation_diff,only:hyper_parameters_to_parameters_b
use mwd_parameters_manipulation_diff,only:hyper_parameters_to_parameters
use mwd_states_manipulation_diff,only:set_states
use mwd_states_manipulation_diff,only:set_states_d
use mwd_states_manipulation_diff,only:set_states_b
use mwd_states_manipulation_diff,only:set_hyper_states
use mwd_states_manipulation_diff,only:get_states_d
use mwd_states_manipulation_diff,only:get_states_b
use mwd_states_manipulation_diff,only:get_states
use mwd_states_manipulation_diff,only:set3d_states_d
use mwd_states_manipulation_diff,only:set3d_states_b
use mwd_states_manipulation_diff,only:set3d_states
use mwd_states_manipulation_diff,only:set1d_states
use mwd_states_manipulation_diff,only:set0d_states
use mwd_states_manipulation_diff,only:normalize_states_d
use mwd_states_manipulation_diff,only:normalize_states_b
use mwd_states_manipulation_diff,only:normalize_states
use mwd_states_manipulation_diff,only:denormalize_states_d
use mwd_states_manipulation_diff,only:denormalize_states_b
use mwd_states_manipulation_diff,only:denormalize_states
use mwd_states_manipulation_diff,only:get_hyper_states_d
use mwd_states_manipulation_diff,only:get_hyper_states_b
use mwd_states_manipulation_diff,only:get_hyper_states
use mwd_states_manipulation_diff,only:set3d_hyper_states
use mwd_states_manipulation_diff,only:set1d_hyper_states
use mwd_states_manipulation_diff,only:set0d_hyper_states
use mwd_states_manipulation_diff,only:hyper_states_to_states_d
use mwd_states_manipulation_diff,only:hyper_states_to_states_b
use mwd_states_manipulation_diff,only:hyper_states_to_states
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
use mwd_mesh,only:meshdt
use mwd_mesh,only:meshdt_initialise
use mwd_setup,only:optimize_setupdt
use mwd_setup,only:setupdt
use mwd_setup,only:optimize_setupdt_initialise
use mwd_setup,only:setupdt_initialise
use mwd_input_data,only:input_datadt
use mwd_input_data,only:input_datadt_initialise
contains
subroutine compute_jobs_d(setup,mesh,input_data,output,output_d,jobs,jobs_d)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(outputdt),intent(inout)::output
type(outputdt_diff),intent(inout)::output_d
real(4),intent(out)::jobs
real(4),intent(out)::jobs_d
end
subroutine compute_jobs_b(setup,mesh,input_data,output,output_b,jobs,jobs_b)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(outputdt),intent(inout)::output
type(outputdt_diff),intent(inout)::output_b
real(4)::jobs
real(4)::jobs_b
end
subroutine compute_jobs(setup,mesh,input_data,output,jobs)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(outputdt),intent(inout)::output
real(4),intent(out)::jobs
end
subroutine compute_jreg_d(setup,mesh,input_data,parameters,parameters_d,parameters_bgd,parameters_bgd_d,states,states_d,states_bgd,states_bgd_d,jreg,jreg_d)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(parametersdt),intent(in)::parameters
type(parametersdt),intent(in)::parameters_d
type(parametersdt),intent(in)::parameters_bgd
type(parametersdt),intent(in)::parameters_bgd_d
type(statesdt),intent(in)::states
type(statesdt),intent(in)::states_d
type(statesdt),intent(in)::states_bgd
type(statesdt),intent(in)::states_bgd_d
real(4),intent(inout)::jreg
real(4),intent(inout)::jreg_d
end
subroutine compute_jreg_b(setup,mesh,input_data,parameters,parameters_b,parameters_bgd,parameters_bgd_b,states,states_b,states_bgd,states_bgd_b,jreg,jreg_b)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(parametersdt),intent(in)::parameters
type(parametersdt)::parameters_b
type(parametersdt),intent(in)::parameters_bgd
type(parametersdt)::parameters_bgd_b
type(statesdt),intent(in)::states
type(statesdt)::states_b
type(statesdt),intent(in)::states_bgd
type(statesdt)::states_bgd_b
real(4),intent(inout)::jreg
real(4),intent(inout)::jreg_b
end
subroutine compute_jreg(setup,mesh,input_data,parameters,parameters_bgd,states,states_bgd,jreg)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(parametersdt),intent(in)::parameters
type(parametersdt),intent(in)::parameters_bgd
type(statesdt),intent(in)::states
type(statesdt),intent(in)::states_bgd
real(4),intent(inout)::jreg
end
subroutine compute_cost_d(setup,mesh,input_data,parameters,parameters_d,parameters_bgd,parameters_bgd_d,states,states_d,states_bgd,states_bgd_d,output,output_d,cost,cost_d)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(parametersdt),intent(inout)::parameters
type(parametersdt),intent(inout)::parameters_d
type(parametersdt),intent(in)::parameters_bgd
type(parametersdt),intent(in)::parameters_bgd_d
type(statesdt),intent(inout)::states
type(statesdt),intent(inout)::states_d
type(statesdt),intent(in)::states_bgd
type(statesdt),intent(in)::states_bgd_d
type(outputdt),intent(inout)::output
type(outputdt_diff),intent(inout)::output_d
real(4),intent(inout)::cost
real(4),intent(inout)::cost_d
end
subroutine compute_cost_b(setup,mesh,input_data,parameters,parameters_b,parameters_bgd,parameters_bgd_b,states,states_b,states_bgd,states_bgd_b,output,output_b,cost,cost_b)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(parametersdt),intent(inout)::parameters
type(parametersdt),intent(inout)::parameters_b
type(parametersdt),intent(in)::parameters_bgd
type(parametersdt)::parameters_bgd_b
type(statesdt),intent(inout)::states
type(statesdt),intent(inout)::states_b
type(statesdt),intent(in)::states_bgd
type(statesdt)::states_bgd_b
type(outputdt),intent(inout)::output
type(outputdt_diff),intent(inout)::output_b
real(4),intent(inout)::cost
real(4),intent(inout)::cost_b
end
subroutine compute_cost(setup,mesh,input_data,parameters,parameters_bgd,states,states_bgd,output,cost)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(parametersdt),intent(inout)::parameters
type(parametersdt),intent(in)::parameters_bgd
type(statesdt),intent(inout)::states
type(statesdt),intent(in)::states_bgd
type(outputdt),intent(inout)::output
real(4),intent(inout)::cost
end
subroutine hyper_compute_cost_d(setup,mesh,input_data,hyper_parameters,hyper_parameters_bgd,hyper_states,hyper_states_bgd,output,output_d,cost,cost_d)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(hyper_parametersdt),intent(in)::hyper_parameters
type(hyper_parametersdt),intent(in)::hyper_parameters_bgd
type(hyper_statesdt),intent(in)::hyper_states
type(hyper_statesdt),intent(in)::hyper_states_bgd
type(outputdt),intent(inout)::output
type(outputdt_diff),intent(inout)::output_d
real(4),intent(inout)::cost
real(4),intent(inout)::cost_d
end
subroutine hyper_compute_cost_b(setup,mesh,input_data,hyper_parameters,hyper_parameters_bgd,hyper_states,hyper_states_bgd,output,output_b,cost,cost_b)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(hyper_parametersdt),intent(in)::hyper_parameters
type(hyper_parametersdt),intent(in)::hyper_parameters_bgd
type(hyper_statesdt),intent(in)::hyper_states
type(hyper_statesdt),intent(in)::hyper_states_bgd
type(outputdt),intent(inout)::output
type(outputdt_diff),intent(inout)::output_b
real(4),intent(inout)::cost
real(4),intent(inout)::cost_b
end
subroutine hyper_compute_cost(setup,mesh,input_data,hyper_parameters,hyper_parameters_bgd,hyper_states,hyper_states_bgd,output,cost)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(hyper_parametersdt),intent(in)::hyper_parameters
type(hyper_parametersdt),intent(in)::hyper_parameters_bgd
type(hyper_statesdt),intent(in)::hyper_states
type(hyper_statesdt),intent(in)::hyper_states_bgd
type(outputdt),intent(inout)::output
real(4),intent(inout)::cost
end
function nse_d(x,y,y_d,res) result(res_d)
real(4),intent(in)::x(:)
real(4),intent(in)::y(:)
real(4),intent(in)::y_d(:)
real(4)::res
real(4)::res_d
end
subroutine nse_b(x,y,y_b,res_b)
real(4),intent(in)::x(:)
real(4),intent(in)::y(:)
real(4)::y_b(:)
real(4)::res_b
end
function nse(x,y) result(res)
real(4),intent(in)::x(:)
real(4),intent(in)::y(:)
real(4)::res
end
subroutine kge_components_d(x,y,y_d,r,r_d,a,a_d,b,b_d)
real(4),intent(in)::x(:)
real(4),intent(in)::y(:)
real(4),intent(in)::y_d(:)
real(4),intent(inout)::r
real(4),intent(inout)::r_d
real(4),intent(inout)::a
real(4),intent(inout)::a_d
real(4),intent(inout)::b
real(4),intent(inout)::b_d
end
subroutine kge_components_b(x,y,y_b,r,r_b,a,a_b,b,b_b)
real(4),intent(in)::x(:)
real(4),intent(in)::y(:)
real(4)::y_b(:)
real(4),intent(inout)::r
real(4),intent(inout)::r_b
real(4),intent(inout)::a
real(4),intent(inout)::a_b
real(4),intent(inout)::b
real(4),intent(inout)::b_b
end
subroutine kge_components(x,y,r,a,b)
real(4),intent(in)::x(:)
real(4),intent(in)::y(:)
real(4),intent(inout)::r
real(4),intent(inout)::a
real(4),intent(inout)::b
end
function kge_d(x,y,y_d,res) result(res_d)
real(4),intent(in)::x(:)
real(4),intent(in)::y(:)
real(4),intent(in)::y_d(:)
real(4)::res
real(4)::res_d
end
subroutine kge_b(x,y,y_b,res_b)
real(4),intent(in)::x(:)
real(4),intent(in)::y(:)
real(4)::y_b(:)
real(4)::res_b
end
function kge(x,y) result(res)
real(4),intent(in)::x(:)
real(4),intent(in)::y(:)
real(4)::res
end
function se_d(x,y,y_d,res) result(res_d)
real(4),intent(in)::x(:)
real(4),intent(in)::y(:)
real(4),intent(in)::y_d(:)
real(4)::res
real(4)::res_d
end
subroutine se_b(x,y,y_b,res_b)
real(4),intent(in)::x(:)
real(4),intent(in)::y(:)
real(4)::y_b(:)
real(4)::res_b
end
function se(x,y) result(res)
real(4),intent(in)::x(:)
real(4),intent(in)::y(:)
real(4)::res
end
function rmse_d(x,y,y_d,res) result(res_d)
real(4),intent(in)::x(:)
real(4),intent(in)::y(:)
real(4),intent(in)::y_d(:)
real(4)::res
real(4)::res_d
end
subroutine rmse_b(x,y,y_b,res_b)
real(4),intent(in)::x(:)
real(4),intent(in)::y(:)
real(4)::y_b(:)
real(4)::res_b
end
function rmse(x,y) result(res)
real(4),intent(in)::x(:)
real(4),intent(in)::y(:)
real(4)::res
end
function logarithmic_d(x,y,y_d,res) result(res_d)
real(4),intent(in)::x(:)
real(4),intent(in)::y(:)
real(4),intent(in)::y_d(:)
real(4)::res
real(4)::res_d
end
subroutine logarithmic_b(x,y,y_b,res_b)
real(4),intent(in)::x(:)
real(4),intent(in)::y(:)
real(4)::y_b(:)
real(4)::res_b
end
function logarithmic(x,y) result(res)
real(4),intent(in)::x(:)
real(4),intent(in)::y(:)
real(4)::res
end
subroutine heap_sort_d(n,arr,arr_d)
integer(4),intent(in)::n
real(4),intent(inout)::arr(1_8:int(n,kind=8))
real(4),intent(inout)::arr_d(1_8:int(n,kind=8))
end
subroutine heap_sort_b(n,arr,arr_b)
integer(4),intent(in)::n
real(4),intent(inout)::arr(1_8:int(n,kind=8))
real(4),intent(inout)::arr_b(1_8:int(n,kind=8))
end
subroutine heap_sort(n,arr)
integer(4),intent(in)::n
real(4),intent(inout)::arr(1_8:int(n,kind=8))
end
function quantile_d(dat,dat_d,p,res) result(res_d)
real(4),intent(in)::dat(:)
real(4),intent(in)::dat_d(:)
real(4),intent(in)::p
real(4)::res
real(4)::res_d
end
subroutine quantile_b(dat,dat_b,p,res_b)
real(4),intent(in)::dat(:)
real(4)::dat_b(:)
real(4),intent(in)::p
real(4)::res_b
end
function quantile(dat,p) result(res)
real(4),intent(in)::dat(:)
real(4),intent(in)::p
real(4)::res
end
subroutine flow_percentile_d(qo,qs,qs_d,p,num,num_d,den)
real(4),intent(in)::qo(:)
real(4),intent(in)::qs(:)
real(4),intent(in)::qs_d(:)
real(4),intent(in)::p
real(4),intent(inout)::num
real(4),intent(inout)::num_d
real(4),intent(inout)::den
end
subroutine flow_percentile_b(qo,qs,qs_b,p,num,num_b,den)
real(4),intent(in)::qo(:)
real(4),intent(in)::qs(:)
real(4)::qs_b(:)
real(4),intent(in)::p
real(4),intent(inout)::num
real(4),intent(inout)::num_b
real(4),intent(inout)::den
end
subroutine flow_percentile(qo,qs,p,num,den)
real(4),intent(in)::qo(:)
real(4),intent(in)::qs(:)
real(4),intent(in)::p
real(4),intent(inout)::num
real(4),intent(inout)::den
end
function signature_d(po,qo,qs,qs_d,mask_event,stype,res) result(res_d)
real(4),intent(in)::po(:)
real(4),intent(in)::qo(:)
real(4),intent(in)::qs(:)
real(4),intent(in)::qs_d(:)
integer(4),intent(in)::mask_event(:)
character(*,1),intent(in)::stype
real(4)::res
real(4)::res_d
end
subroutine signature_b(po,qo,qs,qs_b,mask_event,stype,res_b)
real(4),intent(in)::po(:)
real(4),intent(in)::qo(:)
real(4),intent(in)::qs(:)
real(4)::qs_b(:)
integer(4),intent(in)::mask_event(:)
character(*,1),intent(in)::stype
real(4)::res_b
end
function signature(po,qo,qs,mask_event,stype) result(res)
real(4),intent(in)::po(:)
real(4),intent(in)::qo(:)
real(4),intent(in)::qs(:)
integer(4),intent(in)::mask_event(:)
character(*,1),intent(in)::stype
real(4)::res
end
function distance_correlation_descriptors_d(setup,mesh,input_data,target_control,nbz,parameters_matrix,parameters_matrix_d,penalty_total) result(penalty_total_d)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
character(6_4,1),intent(in)::target_control
integer(4),intent(in)::nbz
real(4),intent(in)::parameters_matrix(1_8:int(mesh%nrow,kind=8),1_8:int(mesh%ncol,kind=8),1_8:int(nbz,kind=8))
real(4),intent(in)::parameters_matrix_d(1_8:int(mesh%nrow,kind=8),1_8:int(mesh%ncol,kind=8),1_8:int(nbz,kind=8))
real(4)::penalty_total
real(4)::penalty_total_d
end
subroutine distance_correlation_descriptors_b(setup,mesh,input_data,target_control,nbz,parameters_matrix,parameters_matrix_b,penalty_total_b)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
character(6_4,1),intent(in)::target_control
integer(4),intent(in)::nbz
real(4),intent(in)::parameters_matrix(1_8:int(mesh%nrow,kind=8),1_8:int(mesh%ncol,kind=8),1_8:int(nbz,kind=8))
real(4)::parameters_matrix_b(1_8:int(mesh%nrow,kind=8),1_8:int(mesh%ncol,kind=8),1_8:int(nbz,kind=8))
real(4)::penalty_total_b
end
function distance_correlation_descriptors(setup,mesh,input_data,target_control,nbz,parameters_matrix) result(penalty_total)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
character(6_4,1),intent(in)::target_control
integer(4),intent(in)::nbz
real(4),intent(in)::parameters_matrix(1_8:int(mesh%nrow,kind=8),1_8:int(mesh%ncol,kind=8),1_8:int(nbz,kind=8))
real(4)::penalty_total
end
function reg_smoothing_d(setup,mesh,optim_arr,matrix,matrix_d,matrix_bgd,rel_to_bgd,res) result(res_d)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
integer(4),intent(in)::optim_arr(:)
real(4),intent(in)::matrix(:,:,:)
real(4),intent(in)::matrix_d(:,:,:)
real(4),intent(in)::matrix_bgd(:,:,:)
logical(4),intent(in)::rel_to_bgd
real(4)::res
real(4)::res_d
end
subroutine reg_smoothing_b(setup,mesh,optim_arr,matrix,matrix_b,matrix_bgd,rel_to_bgd,res_b)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
integer(4),intent(in)::optim_arr(:)
real(4),intent(in)::matrix(:,:,:)
real(4)::matrix_b(:,:,:)
real(4),intent(in)::matrix_bgd(:,:,:)
logical(4),intent(in)::rel_to_bgd
real(4)::res_b
end
function reg_smoothing(setup,mesh,optim_arr,matrix,matrix_bgd,rel_to_bgd) result(res)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
integer(4),intent(in)::optim_arr(:)
real(4),intent(in)::matrix(:,:,:)
real(4),intent(in)::matrix_bgd(:,:,:)
logical(4),intent(in)::rel_to_bgd
real(4)::res
end
function reg_prior_d(setup,optim_arr,matrix,matrix_d,matrix_bgd,res) result(res_d)
type(setupdt),intent(in)::setup
integer(4),intent(in)::optim_arr(:)
real(4),intent(in)::matrix(:,:,:)
real(4),intent(in)::matrix_d(:,:,:)
real(4),intent(in)::matrix_bgd(:,:,:)
real(4)::res
real(4)::res_d
end
subroutine reg_prior_b(setup,optim_arr,matrix,matrix_b,matrix_bgd,res_b)
type(setupdt),intent(in)::setup
integer(4),intent(in)::optim_arr(:)
real(4),intent(in)::matrix(:,:,:)
real(4)::matrix_b(:,:,:)
real(4),intent(in)::matrix_bgd(:,:,:)
real(4)::res_b
end
function reg_prior(setup,optim_arr,matrix,matrix_bgd) result(res)
type(setupdt),intent(in)::setup
integer(4),intent(in)::optim_arr(:)
real(4),intent(in)::matrix(:,:,:)
real(4),intent(in)::matrix_bgd(:,:,:)
real(4)::res
end
end
