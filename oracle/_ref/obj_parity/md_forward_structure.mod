﻿!mod$ v1 sum:8bf08c68998e7808
!need$ eda5fd194b829f52 n mwd_parameters
!need$ 44a770df04028c8f n mwd_output
!need$ 830568814217b360 n md_gr_operator
!need$ 618fc63b1b451b64 n md_vic_operator
!need$ 904794908a6ac3c0 n md_routing_operator
!need$ 76c1396aa4cc0721 n mwd_input_data
!need$ 82a26416841665dd n mwd_setup
!need$ b7e498e07543ba78 n mwd_mesh
!need$ c5f5068eb58aec21 n mwd_states
!need$ 86b12428149ac79a n md_constant
module md_forward_structure
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
use md_gr_operator,only:gr_interception
use md_gr_operator,only:gr_production
use md_gr_operator,only:gr_exchange
use md_gr_operator,only:gr_transfer
use md_vic_operator,only:vic_infiltration
use md_vic_operator,only:vic_vertical_transfer
use md_vic_operator,only:vic_interflow
use md_vic_operator,only:vic_baseflow
use md_vic_operator,only:brooks_and_corey_flow
use md_vic_operator,only:linear_evapotranspiration
use md_routing_operator,only:upstream_discharge
use md_routing_operator,only:linear_routing
contains
subroutine gr_a_forward(setup,mesh,input_data,parameters,states,output)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(parametersdt),intent(in)::parameters
type(statesdt),intent(inout)::states
type(outputdt),intent(inout)::output
end
subroutine gr_b_forward(setup,mesh,input_data,parameters,states,output)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(parametersdt),intent(in)::parameters
type(statesdt),intent(inout)::states
type(outputdt),intent(inout)::output
end
subroutine gr_c_forward(setup,mesh,input_data,parameters,states,output)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(parametersdt),intent(in)::parameters
type(statesdt),intent(inout)::states
type(outputdt),intent(inout)::output
end
subroutine gr_d_forward(setup,mesh,input_data,parameters,states,output)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(parametersdt),intent(in)::parameters
type(statesdt),intent(inout)::states
type(outputdt),intent(inout)::output
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
