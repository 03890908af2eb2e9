﻿!mod$ v1 sum:7ad9fd03fb1bfcc8
!need$ 669a078b745dce26 n mw_forward
!need$ a4c6f86b2bf57c32 n mwd_cost
!need$ 86b12428149ac79a n md_constant
!need$ 82a26416841665dd n mwd_setup
!need$ b7e498e07543ba78 n mwd_mesh
!need$ 76c1396aa4cc0721 n mwd_input_data
!need$ eda5fd194b829f52 n mwd_parameters
!need$ c5f5068eb58aec21 n mwd_states
!need$ 44a770df04028c8f n mwd_output
!need$ 4d57810507808050 n mwd_parameters_manipulation
!need$ ba196de66a1b48a6 n mwd_states_manipulation
module mw_multiple_run
use md_constant,only:sp
use md_constant,only:gnp
use md_constant,only:gns
use mwd_setup,only:setupdt
use mwd_mesh,only:meshdt
use mwd_input_data,only:input_datadt
use mwd_parameters,only:parametersdt
use mwd_states,only:statesdt
use mwd_output,only:outputdt
use mw_forward,only:forward
use mwd_parameters_manipulation,only:get_parameters
use mwd_parameters_manipulation,only:set_parameters
use mwd_states_manipulation,only:get_states
use mwd_states_manipulation,only:set_states
use mwd_cost,only:nse
use mwd_parameters_manipulation,only:mwd_parameters_manipulation$mwd_parameters_manipulation$set3d_parameters=>set3d_parameters
use mwd_states_manipulation,only:mwd_states_manipulation$mwd_states_manipulation$set3d_states=>set3d_states
contains
subroutine wait_bar_multiple_run(iter,niter)
integer(4),intent(in)::iter
integer(4),intent(in)::niter
end
subroutine set_sample_to_parameters_states(mesh,parameters,states,ind_parameters_states,sample_arr)
type(meshdt),intent(in)::mesh
type(parametersdt),intent(inout)::parameters
type(statesdt),intent(inout)::states
integer(4),intent(in)::ind_parameters_states(:)
real(4),intent(in)::sample_arr(:)
end
subroutine compute_multiple_run(setup,mesh,input_data,parameters,states,output,sample,ind_parameters_states,res_cost,res_qsim)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(parametersdt),intent(inout)::parameters
type(statesdt),intent(inout)::states
type(outputdt),intent(inout)::output
real(4),intent(in)::sample(:,:)
integer(4),intent(in)::ind_parameters_states(:)
real(4),intent(inout)::res_cost(:)
real(4),intent(inout)::res_qsim(:,:,:)
end
end
