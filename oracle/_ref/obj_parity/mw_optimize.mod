﻿!mod$ v1 sum:5646c8d7b79c7ea8
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
module mw_optimize
use md_constant,only:sp
use md_constant,only:dp
use md_constant,only:lchar
use md_constant,only:gnp
use md_constant,only:gns
use mwd_setup,only:setupdt
use mwd_mesh,only:meshdt
use mwd_input_data,only:input_datadt
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
use mw_forward,only:forward
use mw_forward,only:forward_b
use mw_forward,only:hyper_forward
use mw_forward,only:hyper_forward_b
use mwd_parameters_manipulation,only:get_parameters
use mwd_parameters_manipulation,only:set_parameters
use mwd_parameters_manipulation,only:get_hyper_parameters
use mwd_parameters_manipulation,only:set_hyper_parameters
use mwd_parameters_manipulation,only:hyper_parameters_to_parameters
use mwd_parameters_manipulation,only:normalize_parameters
use mwd_states_manipulation,only:get_states
use mwd_states_manipulation,only:set_states
use mwd_states_manipulation,only:get_hyper_states
use mwd_states_manipulation,only:set_hyper_states
use mwd_states_manipulation,only:hyper_states_to_states
use mwd_states_manipulation,only:normalize_states
use mwd_parameters_manipulation,only:mwd_parameters_manipulation$mwd_parameters_manipulation$set0d_hyper_parameters=>set0d_hyper_parameters
use mwd_parameters_manipulation,only:mwd_parameters_manipulation$mwd_parameters_manipulation$set3d_hyper_parameters=>set3d_hyper_parameters
use mwd_parameters_manipulation,only:mwd_parameters_manipulation$mwd_parameters_manipulation$set3d_parameters=>set3d_parameters
use mwd_states_manipulation,only:mwd_states_manipulation$mwd_states_manipulation$set0d_hyper_states=>set0d_hyper_states
use mwd_states_manipulation,only:mwd_states_manipulation$mwd_states_manipulation$set3d_hyper_states=>set3d_hyper_states
use mwd_states_manipulation,only:mwd_states_manipulation$mwd_states_manipulation$set3d_states=>set3d_states
private::bounds_initialise_sbs
private::var_to_control_sbs
private::control_to_var_sbs
private::transformation_sbs
private::inv_transformation_sbs
private::var_to_control_lbfgsb
private::control_to_var_lbfgsb
private::normalize_descriptor_hyper_lbfgsb
private::denormalize_descriptor_hyper_lbfgsb
private::problem_initialise_hyper_lbfgsb
private::var_to_control_hyper_lbfgsb
private::control_to_var_hyper_lbfgsb
contains
subroutine optimize_sbs(setup,mesh,input_data,parameters,states,output)
type(setupdt),intent(inout)::setup
type(meshdt),intent(inout)::mesh
type(input_datadt),intent(inout)::input_data
type(parametersdt),intent(inout)::parameters
type(statesdt),intent(inout)::states
type(outputdt),intent(inout)::output
end
subroutine bounds_initialise_sbs(n,setup,l,u)
integer(4),intent(in)::n
type(setupdt),intent(in)::setup
real(4),intent(inout)::l(1_8:int(n,kind=8))
real(4),intent(inout)::u(1_8:int(n,kind=8))
end
subroutine var_to_control_sbs(n,setup,mesh,parameters,states,x)
integer(4),intent(in)::n
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(parametersdt),intent(in)::parameters
type(statesdt),intent(in)::states
real(4),intent(inout)::x(1_8:int(n,kind=8))
end
subroutine control_to_var_sbs(n,setup,mesh,parameters,states,x)
integer(4),intent(in)::n
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(parametersdt),intent(inout)::parameters
type(statesdt),intent(inout)::states
real(4),intent(in)::x(1_8:int(n,kind=8))
end
subroutine transformation_sbs(n,x,l,u,x_t)
integer(4),intent(in)::n
real(4),intent(in)::x(1_8:int(n,kind=8))
real(4),intent(in)::l(1_8:int(n,kind=8))
real(4),intent(in)::u(1_8:int(n,kind=8))
real(4),intent(inout)::x_t(1_8:int(n,kind=8))
end
subroutine inv_transformation_sbs(n,x,l,u,x_t)
integer(4),intent(in)::n
real(4),intent(inout)::x(1_8:int(n,kind=8))
real(4),intent(in)::l(1_8:int(n,kind=8))
real(4),intent(in)::u(1_8:int(n,kind=8))
real(4),intent(in)::x_t(1_8:int(n,kind=8))
end
subroutine optimize_lbfgsb(setup,mesh,input_data,parameters,states,output)
type(setupdt),intent(inout)::setup
type(meshdt),intent(inout)::mesh
type(input_datadt),intent(inout)::input_data
type(parametersdt),intent(inout)::parameters
type(statesdt),intent(inout)::states
type(outputdt),intent(inout)::output
end
subroutine var_to_control_lbfgsb(n,setup,mesh,parameters,states,x)
integer(4),intent(in)::n
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(parametersdt),intent(in)::parameters
type(statesdt),intent(in)::states
real(8),intent(inout)::x(1_8:int(n,kind=8))
end
subroutine control_to_var_lbfgsb(n,setup,mesh,parameters,states,x)
integer(4),intent(in)::n
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(parametersdt),intent(inout)::parameters
type(statesdt),intent(inout)::states
real(8),intent(in)::x(1_8:int(n,kind=8))
end
subroutine optimize_hyper_lbfgsb(setup,mesh,input_data,parameters,states,output)
type(setupdt),intent(inout)::setup
type(meshdt),intent(inout)::mesh
type(input_datadt),intent(inout)::input_data
type(parametersdt),intent(inout)::parameters
type(statesdt),intent(inout)::states
type(outputdt),intent(inout)::output
end
subroutine normalize_descriptor_hyper_lbfgsb(setup,input_data,min_descriptor,max_descriptor)
type(setupdt),intent(in)::setup
type(input_datadt),intent(inout)::input_data
real(4)::min_descriptor(1_8:int(setup%nd,kind=8))
real(4)::max_descriptor(1_8:int(setup%nd,kind=8))
end
subroutine denormalize_descriptor_hyper_lbfgsb(setup,input_data,min_descriptor,max_descriptor)
type(setupdt),intent(in)::setup
type(input_datadt),intent(inout)::input_data
real(4)::min_descriptor(1_8:int(setup%nd,kind=8))
real(4)::max_descriptor(1_8:int(setup%nd,kind=8))
end
subroutine problem_initialise_hyper_lbfgsb(n,setup,mesh,parameters,states,hyper_parameters,hyper_states,nbd,l,u)
integer(4),intent(in)::n
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(parametersdt),intent(in)::parameters
type(statesdt),intent(in)::states
type(hyper_parametersdt),intent(inout)::hyper_parameters
type(hyper_statesdt),intent(inout)::hyper_states
integer(4),intent(inout)::nbd(1_8:int(n,kind=8))
real(8),intent(inout)::l(1_8:int(n,kind=8))
real(8),intent(inout)::u(1_8:int(n,kind=8))
end
subroutine var_to_control_hyper_lbfgsb(n,setup,hyper_parameters,hyper_states,x)
integer(4),intent(in)::n
type(setupdt),intent(in)::setup
type(hyper_parametersdt),intent(in)::hyper_parameters
type(hyper_statesdt),intent(in)::hyper_states
real(8),intent(inout)::x(1_8:int(n,kind=8))
end
subroutine control_to_var_hyper_lbfgsb(n,setup,hyper_parameters,hyper_states,x)
integer(4),intent(in)::n
type(setupdt),intent(in)::setup
type(hyper_parametersdt),intent(inout)::hyper_parameters
type(hyper_statesdt),intent(inout)::hyper_states
real(8),intent(in)::x(1_8:int(n,kind=8))
end
end
