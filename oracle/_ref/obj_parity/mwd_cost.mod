﻿!mod$ v1 sum:a4c6f86b2bf57c32
!need$ 44a770df04028c8f n mwd_output
!need$ 4d57810507808050 n mwd_parameters_manipulation
!need$ ba196de66a1b48a6 n mwd_states_manipulation
!need$ b7e498e07543ba78 n mwd_mesh
!need$ eda5fd194b829f52 n mwd_parameters
!need$ 86b12428149ac79a n md_constant
!need$ 82a26416841665dd n mwd_setup
!need$ 76c1396aa4cc0721 n mwd_input_data
!need$ c5f5068eb58aec21 n mwd_states
module mwd_cost
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
contains
subroutine compute_jobs(setup,mesh,input_data,output,jobs)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(outputdt),intent(inout)::output
real(4),intent(out)::jobs
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
function nse(x,y) result(res)
real(4),intent(in)::x(:)
real(4),intent(in)::y(:)
real(4)::res
end
subroutine kge_components(x,y,r,a,b)
real(4),intent(in)::x(:)
real(4),intent(in)::y(:)
real(4),intent(inout)::r
real(4),intent(inout)::a
real(4),intent(inout)::b
end
function kge(x,y) result(res)
real(4),intent(in)::x(:)
real(4),intent(in)::y(:)
real(4)::res
end
function se(x,y) result(res)
real(4),intent(in)::x(:)
real(4),intent(in)::y(:)
real(4)::res
end
function rmse(x,y) result(res)
real(4),intent(in)::x(:)
real(4),intent(in)::y(:)
real(4)::res
end
function logarithmic(x,y) result(res)
real(4),intent(in)::x(:)
real(4),intent(in)::y(:)
real(4)::res
end
subroutine heap_sort(n,arr)
integer(4),intent(in)::n
real(4),intent(inout)::arr(1_8:int(n,kind=8))
end
function quantile(dat,p) result(res)
real(4),intent(in)::dat(:)
real(4),intent(in)::p
real(4)::res
end
subroutine flow_percentile(qo,qs,p,num,den)
real(4),intent(in)::qo(:)
real(4),intent(in)::qs(:)
real(4),intent(in)::p
real(4),intent(inout)::num
real(4),intent(inout)::den
end
function signature(po,qo,qs,mask_event,stype) result(res)
real(4),intent(in)::po(:)
real(4),intent(in)::qo(:)
real(4),intent(in)::qs(:)
integer(4),intent(in)::mask_event(:)
character(*,1),intent(in)::stype
real(4)::res
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
function reg_smoothing(setup,mesh,optim_arr,matrix,matrix_bgd,rel_to_bgd) result(res)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
integer(4),intent(in)::optim_arr(:)
real(4),intent(in)::matrix(:,:,:)
real(4),intent(in)::matrix_bgd(:,:,:)
logical(4),intent(in)::rel_to_bgd
real(4)::res
end
function reg_prior(setup,optim_arr,matrix,matrix_bgd) result(res)
type(setupdt),intent(in)::setup
integer(4),intent(in)::optim_arr(:)
real(4),intent(in)::matrix(:,:,:)
real(4),intent(in)::matrix_bgd(:,:,:)
real(4)::res
end
end
