﻿!mod$ v1 sum:669a078b745dce26
!need$ eda5fd194b829f52 n mwd_parameters
!need$ 44a770df04028c8f n mwd_output
!need$ 76c1396aa4cc0721 n mwd_input_data
!need$ 86b12428149ac79a n md_constant
!need$ 82a26416841665dd n mwd_setup
!need$ b7e498e07543ba78 n mwd_mesh
!need$ c5f5068eb58aec21 n mwd_states
module mw_forward
use md_constant,only:sp
use md_constant,only:dp
use mwd_setup,only:setupdt
use mwd_mesh,only:meshdt
use mwd_input_data,only:input_datadt
use mwd_parameters,only:parametersdt
use mwd_parameters,only:hyper_parametersdt
use mwd_states,only:statesdt
use mwd_states,only:hyper_statesdt
use mwd_output,only:outputdt
contains
subroutine forward(setup,mesh,input_data,parameters,parameters_bgd,states,states_bgd,output,cost)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(parametersdt),intent(inout)::parameters
type(parametersdt),intent(inout)::parameters_bgd
type(statesdt),intent(inout)::states
type(statesdt),intent(inout)::states_bgd
type(outputdt),intent(inout)::output
real(4),intent(inout)::cost
end
subroutine forward_b(setup,mesh,input_data,parameters,parameters_b,parameters_bgd,parameters_bgd_b,states,states_b,states_bgd,states_bgd_b,output,output_b,cost,cost_b)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(parametersdt),intent(inout)::parameters
type(parametersdt),intent(inout)::parameters_b
type(parametersdt),intent(inout)::parameters_bgd
type(parametersdt),intent(inout)::parameters_bgd_b
type(statesdt),intent(inout)::states
type(statesdt),intent(inout)::states_b
type(statesdt),intent(inout)::states_bgd
type(statesdt),intent(inout)::states_bgd_b
type(outputdt),intent(inout)::output
type(outputdt),intent(inout)::output_b
real(4),intent(inout)::cost
real(4),intent(inout)::cost_b
end
subroutine forward_d(setup,mesh,input_data,parameters,parameters_d,parameters_bgd,parameters_bgd_d,states,states_d,states_bgd,states_bgd_d,output,output_d,cost,cost_d)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(parametersdt),intent(inout)::parameters
type(parametersdt),intent(inout)::parameters_d
type(parametersdt),intent(inout)::parameters_bgd
type(parametersdt),intent(inout)::parameters_bgd_d
type(statesdt),intent(inout)::states
type(statesdt),intent(inout)::states_d
type(statesdt),intent(inout)::states_bgd
type(statesdt),intent(inout)::states_bgd_d
type(outputdt),intent(inout)::output
type(outputdt),intent(inout)::output_d
real(4),intent(inout)::cost
real(4),intent(inout)::cost_d
end
subroutine hyper_forward(setup,mesh,input_data,parameters,hyper_parameters,hyper_parameters_bgd,states,hyper_states,hyper_states_bgd,output,cost)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(parametersdt),intent(inout)::parameters
type(hyper_parametersdt),intent(inout)::hyper_parameters
type(hyper_parametersdt),intent(inout)::hyper_parameters_bgd
type(statesdt),intent(inout)::states
type(hyper_statesdt),intent(inout)::hyper_states
type(hyper_statesdt),intent(inout)::hyper_states_bgd
type(outputdt),intent(inout)::output
real(4),intent(inout)::cost
end
subroutine hyper_forward_b(setup,mesh,input_data,parameters,parameters_b,hyper_parameters,hyper_parameters_b,hyper_parameters_bgd,states,states_b,hyper_states,hyper_states_b,hyper_states_bgd,output,output_b,cost,cost_b)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(parametersdt),intent(inout)::parameters
type(parametersdt),intent(inout)::parameters_b
type(hyper_parametersdt),intent(inout)::hyper_parameters
type(hyper_parametersdt),intent(inout)::hyper_parameters_b
type(hyper_parametersdt),intent(inout)::hyper_parameters_bgd
type(statesdt),intent(inout)::states
type(statesdt),intent(inout)::states_b
type(hyper_statesdt),intent(inout)::hyper_states
type(hyper_statesdt),intent(inout)::hyper_states_b
type(hyper_statesdt),intent(inout)::hyper_states_bgd
type(outputdt),intent(inout)::output
type(outputdt),intent(inout)::output_b
real(4),intent(inout)::cost
real(4),intent(inout)::cost_b
end
subroutine hyper_forward_d(setup,mesh,input_data,parameters,parameters_d,hyper_parameters,hyper_parameters_d,hyper_parameters_bgd,states,states_d,hyper_states,hyper_states_d,hyper_states_bgd,output,output_d,cost,cost_d)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(parametersdt),intent(inout)::parameters
type(parametersdt),intent(inout)::parameters_d
type(hyper_parametersdt),intent(inout)::hyper_parameters
type(hyper_parametersdt),intent(inout)::hyper_parameters_d
type(hyper_parametersdt),intent(inout)::hyper_parameters_bgd
type(statesdt),intent(inout)::states
type(statesdt),intent(inout)::states_d
type(hyper_statesdt),intent(inout)::hyper_states
type(hyper_statesdt),intent(inout)::hyper_states_d
type(hyper_statesdt),intent(inout)::hyper_states_bgd
type(outputdt),intent(inout)::output
type(outputdt),intent(inout)::output_d
real(4),intent(inout)::cost
real(4),intent(inout)::cost_d
end
end
