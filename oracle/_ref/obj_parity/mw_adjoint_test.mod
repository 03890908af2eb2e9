﻿!mod$ v1 sum:9281e4a8cdd54cfa
!need$ 669a078b745dce26 n mw_forward
!need$ 4d57810507808050 n mwd_parameters_manipulation
!need$ ba196de66a1b48a6 n mwd_states_manipulation
!need$ 44a770df04028c8f n mwd_output
!need$ b7e498e07543ba78 n mwd_mesh
!need$ eda5fd194b829f52 n mwd_parameters
!need$ 86b12428149ac79a n md_constant
!need$ 82a26416841665dd n mwd_setup
!need$ 76c1396aa4cc0721 n mwd_input_data
!need$ c5f5068eb58aec21 n mwd_states
module mw_adjoint_test
use md_constant,only:sp
use md_constant,only:gnp
use md_constant,only:gns
use mwd_setup,only:setupdt
use mwd_mesh,only:meshdt
use mwd_input_data,only:input_datadt
use mwd_parameters,only:parametersdt
use mwd_parameters,only:parametersdt_initialise
use mwd_states,only:statesdt
use mwd_states,only:statesdt_initialise
use mwd_output,only:outputdt
use mwd_output,only:outputdt_initialise
use mw_forward,only:forward
use mw_forward,only:forward_b
use mw_forward,only:forward_d
use mwd_parameters_manipulation,only:get_parameters
use mwd_parameters_manipulation,only:set_parameters
use mwd_states_manipulation,only:get_states
use mwd_states_manipulation,only:set_states
use mwd_parameters_manipulation,only:mwd_parameters_manipulation$mwd_parameters_manipulation$set0d_parameters=>set0d_parameters
use mwd_parameters_manipulation,only:mwd_parameters_manipulation$mwd_parameters_manipulation$set3d_parameters=>set3d_parameters
use mwd_states_manipulation,only:mwd_states_manipulation$mwd_states_manipulation$set0d_states=>set0d_states
contains
subroutine scalar_product_test(setup,mesh,input_data,parameters,states,output)
type(setupdt),intent(inout)::setup
type(meshdt),intent(inout)::mesh
type(input_datadt),intent(inout)::input_data
type(parametersdt),intent(inout)::parameters
type(statesdt),intent(inout)::states
type(outputdt),intent(inout)::output
end
subroutine gradient_test(setup,mesh,input_data,parameters,states,output)
type(setupdt),intent(inout)::setup
type(meshdt),intent(inout)::mesh
type(input_datadt),intent(inout)::input_data
type(parametersdt),intent(inout)::parameters
type(statesdt),intent(inout)::states
type(outputdt),intent(inout)::output
end
end
