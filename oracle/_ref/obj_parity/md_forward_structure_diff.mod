﻿!mod$ v1 sum:e84e5c7a4dc214d8
!need$ 54215331fdee0196 n md_gr_operator_diff
!need$ c83657007410df9a n md_vic_operator_diff
!need$ 9a829c5973690d48 n md_routing_operator_diff
!need$ 86b12428149ac79a n md_constant
!need$ 82a26416841665dd n mwd_setup
!need$ b7e498e07543ba78 n mwd_mesh
!need$ 76c1396aa4cc0721 n mwd_input_data
!need$ b688d5c5e1bddd84 n mwd_parameters_diff
!need$ 23b6a22028cfb5af n mwd_states_diff
!need$ 3d7fc12a1465e543 n mwd_output_diff
module md_forward_structure_diff
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
use md_gr_operator_diff,only:gr_interception_d
use md_gr_operator_diff,only:gr_interception_b
use md_gr_operator_diff,only:gr_interception
use md_gr_operator_diff,only:gr_production_d
use md_gr_operator_diff,only:gr_production_b
use md_gr_operator_diff,only:gr_production
use md_gr_operator_diff,only:gr_exchange_d
use md_gr_operator_diff,only:gr_exchange_b
use md_gr_operator_diff,only:gr_exchange
use md_gr_operator_diff,only:gr_transfer_d
use md_gr_operator_diff,only:gr_transfer_b
use md_gr_operator_diff,only:gr_transfer
use md_routing_operator_diff,only:upstream_discharge_d
use md_routing_operator_diff,only:upstream_discharge_b
use md_routing_operator_diff,only:upstream_discharge
use md_routing_operator_diff,only:linear_routing_d
use md_routing_operator_diff,only:linear_routing_b
use md_routing_operator_diff,only:linear_routing
use md_vic_operator_diff,only:vic_infiltration_d
use md_vic_operator_diff,only:vic_infiltration_b
use md_vic_operator_diff,only:vic_infiltration
use md_vic_operator_diff,only:vic_vertical_transfer_d
use md_vic_operator_diff,only:vic_vertical_transfer_b
use md_vic_operator_diff,only:vic_vertical_transfer
use md_vic_operator_diff,only:vic_interflow_d
use md_vic_operator_diff,only:vic_interflow_b
use md_vic_operator_diff,only:vic_interflow
use md_vic_operator_diff,only:vic_baseflow_d
use md_vic_operator_diff,only:vic_baseflow_b
use md_vic_operator_diff,only:vic_baseflow
use md_vic_operator_diff,only:brooks_and_corey_flow_d
use md_vic_operator_diff,only:brooks_and_corey_flow_b
use md_vic_operator_diff,only:brooks_and_corey_flow
use md_vic_operator_diff,only:linear_evapotranspiration_d
use md_vic_operator_diff,only:linear_evapotranspiration_b
use md_vic_operator_diff,only:linear_evapotranspiration
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
subroutine gr_a_forward_d(setup,mesh,input_data,parameters,parameters_d,states,states_d,output,output_d)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(parametersdt),intent(in)::parameters
type(parametersdt),intent(in)::parameters_d
type(statesdt),intent(inout)::states
type(statesdt),intent(inout)::states_d
type(outputdt),intent(inout)::output
type(outputdt_diff),intent(inout)::output_d
end
subroutine gr_a_forward_b(setup,mesh,input_data,parameters,parameters_b,states,states_b,output,output_b)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(parametersdt),intent(in)::parameters
type(parametersdt)::parameters_b
type(statesdt),intent(inout)::states
type(statesdt),intent(inout)::states_b
type(outputdt),intent(inout)::output
type(outputdt_diff),intent(inout)::output_b
end
subroutine gr_a_forward(setup,mesh,input_data,parameters,states,output)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(parametersdt),intent(in)::parameters
type(statesdt),intent(inout)::states
type(outputdt),intent(inout)::output
end
subroutine gr_b_forward_d(setup,mesh,input_data,parameters,parameters_d,states,states_d,output,output_d)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(parametersdt),intent(in)::parameters
type(parametersdt),intent(in)::parameters_d
type(statesdt),intent(inout)::states
type(statesdt),intent(inout)::states_d
type(outputdt),intent(inout)::output
type(outputdt_diff),intent(inout)::output_d
end
subroutine gr_b_forward_b(setup,mesh,input_data,parameters,parameters_b,states,states_b,output,output_b)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(parametersdt),intent(in)::parameters
type(parametersdt)::parameters_b
type(statesdt),intent(inout)::states
type(statesdt),intent(inout)::states_b
type(outputdt),intent(inout)::output
type(outputdt_diff),intent(inout)::output_b
end
subroutine gr_b_forward(setup,mesh,input_data,parameters,states,output)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(parametersdt),intent(in)::parameters
type(statesdt),intent(inout)::states
type(outputdt),intent(inout)::output
end
subroutine gr_c_forward_d(setup,mesh,input_data,parameters,parameters_d,states,states_d,output,output_d)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(parametersdt),intent(in)::parameters
type(parametersdt),intent(in)::parameters_d
type(statesdt),intent(inout)::states
type(statesdt),intent(inout)::states_d
type(outputdt),intent(inout)::output
type(outputdt_diff),intent(inout)::output_d
end
subroutine gr_c_forward_b(setup,mesh,input_data,parameters,parameters_b,states,states_b,output,output_b)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(parametersdt),intent(in)::parameters
type(parametersdt)::parameters_b
type(statesdt),intent(inout)::states
type(statesdt),intent(inout)::states_b
type(outputdt),intent(inout)::output
type(outputdt_diff),intent(inout)::output_b
end
subroutine gr_c_forward(setup,mesh,input_data,parameters,states,output)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(parametersdt),intent(in)::parameters
type(statesdt),intent(inout)::states
type(outputdt),intent(inout)::output
end
subroutine gr_d_forward_d(setup,mesh,input_data,parameters,parameters_d,states,states_d,output,output_d)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(parametersdt),intent(in)::parameters
type(parametersdt),intent(in)::parameters_d
type(statesdt),intent(inout)::states
type(statesdt),intent(inout)::states_d
type(outputdt),intent(inout)::output
type(outputdt_diff),intent(inout)::output_d
end
subroutine gr_d_forward_b(setup,mesh,input_data,parameters,parameters_b,states,states_b,output,output_b)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(parametersdt),intent(in)::parameters
type(parametersdt)::parameters_b
type(statesdt),intent(inout)::states
type(statesdt),intent(inout)::states_b
type(outputdt),intent(inout)::output
type(outputdt_diff),intent(inout)::output_b
end
subroutine gr_d_forward(setup,mesh,input_data,parameters,states,output)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(parametersdt),intent(in)::parameters
type(statesdt),intent(inout)::states
type(outputdt),intent(inout)::output
end
subroutine vic_a_forward_d(setup,mesh,input_data,parameters,parameters_d,states,states_d,output,output_d)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(parametersdt),intent(in)::parameters
type(parametersdt),intent(in)::parameters_d
type(statesdt),intent(inout)::states
type(statesdt),intent(inout)::states_d
type(outputdt),intent(inout)::output
type(outputdt_diff),intent(inout)::output_d
end
subroutine vic_a_forward_b(setup,mesh,input_data,parameters,parameters_b,states,states_b,output,output_b)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(parametersdt),intent(in)::parameters
type(parametersdt)::parameters_b
type(statesdt),intent(inout)::states
type(statesdt),intent(inout)::states_b
type(outputdt),intent(inout)::output
type(outputdt_diff),intent(inout)::output_b
end
subroutine vic_a_forward(setup,mesh,input_data,parameters,states,output)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(parametersdt),intent(in)::parameters
type(statesdt),intent(inout)::states
type(outputdt),intent(inout)::output
end
end
