﻿!mod$ v1 sum:c83657007410df9a
!need$ 86b12428149ac79a n md_constant
module md_vic_operator_diff
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
contains
subroutine vic_infiltration_d(prcp,cusl1,cusl1_d,cusl2,cusl2_d,b,b_d,husl1,husl1_d,husl2,husl2_d,runoff,runoff_d)
real(4),intent(in)::prcp
real(4),intent(in)::cusl1
real(4),intent(in)::cusl1_d
real(4),intent(in)::cusl2
real(4),intent(in)::cusl2_d
real(4),intent(in)::b
real(4),intent(in)::b_d
real(4),intent(inout)::husl1
real(4),intent(inout)::husl1_d
real(4),intent(inout)::husl2
real(4),intent(inout)::husl2_d
real(4),intent(out)::runoff
real(4),intent(out)::runoff_d
end
subroutine vic_infiltration_b(prcp,cusl1,cusl1_b,cusl2,cusl2_b,b,b_b,husl1,husl1_b,husl2,husl2_b,runoff,runoff_b)
real(4),intent(in)::prcp
real(4),intent(in)::cusl1
real(4)::cusl1_b
real(4),intent(in)::cusl2
real(4)::cusl2_b
real(4),intent(in)::b
real(4)::b_b
real(4),intent(inout)::husl1
real(4),intent(inout)::husl1_b
real(4),intent(inout)::husl2
real(4),intent(inout)::husl2_b
real(4)::runoff
real(4)::runoff_b
end
subroutine vic_infiltration(prcp,cusl1,cusl2,b,husl1,husl2,runoff)
real(4),intent(in)::prcp
real(4),intent(in)::cusl1
real(4),intent(in)::cusl2
real(4),intent(in)::b
real(4),intent(inout)::husl1
real(4),intent(inout)::husl2
real(4),intent(out)::runoff
end
subroutine vic_vertical_transfer_d(pet,cusl1,cusl1_d,cusl2,cusl2_d,clsl,clsl_d,ks,ks_d,husl1,husl1_d,husl2,husl2_d,hlsl,hlsl_d)
real(4),intent(in)::pet
real(4),intent(in)::cusl1
real(4),intent(in)::cusl1_d
real(4),intent(in)::cusl2
real(4),intent(in)::cusl2_d
real(4),intent(in)::clsl
real(4),intent(in)::clsl_d
real(4),intent(in)::ks
real(4),intent(in)::ks_d
real(4),intent(inout)::husl1
real(4),intent(inout)::husl1_d
real(4),intent(inout)::husl2
real(4),intent(inout)::husl2_d
real(4),intent(inout)::hlsl
real(4),intent(inout)::hlsl_d
end
subroutine vic_vertical_transfer_b(pet,cusl1,cusl1_b,cusl2,cusl2_b,clsl,clsl_b,ks,ks_b,husl1,husl1_b,husl2,husl2_b,hlsl,hlsl_b)
real(4),intent(in)::pet
real(4),intent(in)::cusl1
real(4)::cusl1_b
real(4),intent(in)::cusl2
real(4)::cusl2_b
real(4),intent(in)::clsl
real(4)::clsl_b
real(4),intent(in)::ks
real(4)::ks_b
real(4),intent(inout)::husl1
real(4),intent(inout)::husl1_b
real(4),intent(inout)::husl2
real(4),intent(inout)::husl2_b
real(4),intent(inout)::hlsl
real(4),intent(inout)::hlsl_b
end
subroutine vic_vertical_transfer(pet,cusl1,cusl2,clsl,ks,husl1,husl2,hlsl)
real(4),intent(in)::pet
real(4),intent(in)::cusl1
real(4),intent(in)::cusl2
real(4),intent(in)::clsl
real(4),intent(in)::ks
real(4),intent(inout)::husl1
real(4),intent(inout)::husl2
real(4),intent(inout)::hlsl
end
subroutine vic_interflow_d(n,cusl2,cusl2_d,husl2,husl2_d,qi,qi_d)
real(4),intent(in)::n
real(4),intent(in)::cusl2
real(4),intent(in)::cusl2_d
real(4),intent(inout)::husl2
real(4),intent(inout)::husl2_d
real(4),intent(out)::qi
real(4),intent(out)::qi_d
end
subroutine vic_interflow_b(n,cusl2,cusl2_b,husl2,husl2_b,qi,qi_b)
real(4),intent(in)::n
real(4),intent(in)::cusl2
real(4)::cusl2_b
real(4),intent(inout)::husl2
real(4),intent(inout)::husl2_b
real(4)::qi
real(4)::qi_b
end
subroutine vic_interflow(n,cusl2,husl2,qi)
real(4),intent(in)::n
real(4),intent(in)::cusl2
real(4),intent(inout)::husl2
real(4),intent(out)::qi
end
subroutine vic_baseflow_d(clsl,clsl_d,ds,ds_d,dsm,dsm_d,ws,ws_d,hlsl,hlsl_d,qb,qb_d)
real(4),intent(in)::clsl
real(4),intent(in)::clsl_d
real(4),intent(in)::ds
real(4),intent(in)::ds_d
real(4),intent(in)::dsm
real(4),intent(in)::dsm_d
real(4),intent(in)::ws
real(4),intent(in)::ws_d
real(4),intent(inout)::hlsl
real(4),intent(inout)::hlsl_d
real(4),intent(out)::qb
real(4),intent(out)::qb_d
end
subroutine vic_baseflow_b(clsl,clsl_b,ds,ds_b,dsm,dsm_b,ws,ws_b,hlsl,hlsl_b,qb,qb_b)
real(4),intent(in)::clsl
real(4)::clsl_b
real(4),intent(in)::ds
real(4)::ds_b
real(4),intent(in)::dsm
real(4)::dsm_b
real(4),intent(in)::ws
real(4)::ws_b
real(4),intent(inout)::hlsl
real(4),intent(inout)::hlsl_b
real(4)::qb
real(4)::qb_b
end
subroutine vic_baseflow(clsl,ds,dsm,ws,hlsl,qb)
real(4),intent(in)::clsl
real(4),intent(in)::ds
real(4),intent(in)::dsm
real(4),intent(in)::ws
real(4),intent(inout)::hlsl
real(4),intent(out)::qb
end
subroutine brooks_and_corey_flow_d(ks,ks_d,residual,porosity,lambda,c_upper,c_upper_d,c_lower,c_lower_d,h_upper,h_upper_d,h_lower,h_lower_d,flow,flow_d)
real(4),intent(in)::ks
real(4),intent(in)::ks_d
real(4),intent(in)::residual
real(4),intent(in)::porosity
real(4),intent(in)::lambda
real(4),intent(in)::c_upper
real(4),intent(in)::c_upper_d
real(4),intent(in)::c_lower
real(4),intent(in)::c_lower_d
real(4),intent(in)::h_upper
real(4),intent(in)::h_upper_d
real(4),intent(in)::h_lower
real(4),intent(in)::h_lower_d
real(4),intent(out)::flow
real(4),intent(out)::flow_d
end
subroutine brooks_and_corey_flow_b(ks,ks_b,residual,porosity,lambda,c_upper,c_upper_b,c_lower,c_lower_b,h_upper,h_upper_b,h_lower,h_lower_b,flow,flow_b)
real(4),intent(in)::ks
real(4)::ks_b
real(4),intent(in)::residual
real(4),intent(in)::porosity
real(4),intent(in)::lambda
real(4),intent(in)::c_upper
real(4)::c_upper_b
real(4),intent(in)::c_lower
real(4)::c_lower_b
real(4),intent(in)::h_upper
real(4)::h_upper_b
real(4),intent(in)::h_lower
real(4)::h_lower_b
real(4)::flow
real(4)::flow_b
end
subroutine brooks_and_corey_flow(ks,residual,porosity,lambda,c_upper,c_lower,h_upper,h_lower,flow)
real(4),intent(in)::ks
real(4),intent(in)::residual
real(4),intent(in)::porosity
real(4),intent(in)::lambda
real(4),intent(in)::c_upper
real(4),intent(in)::c_lower
real(4),intent(in)::h_upper
real(4),intent(in)::h_lower
real(4),intent(out)::flow
end
subroutine linear_evapotranspiration_d(e,e_d,c,c_d,h,h_d,flow,flow_d)
real(4),intent(in)::e
real(4),intent(in)::e_d
real(4),intent(in)::c
real(4),intent(in)::c_d
real(4),intent(in)::h
real(4),intent(in)::h_d
real(4),intent(out)::flow
real(4),intent(out)::flow_d
end
subroutine linear_evapotranspiration_b(e,e_b,c,c_b,h,h_b,flow,flow_b)
real(4),intent(in)::e
real(4)::e_b
real(4),intent(in)::c
real(4)::c_b
real(4),intent(in)::h
real(4)::h_b
real(4)::flow
real(4)::flow_b
end
subroutine linear_evapotranspiration(e,c,h,flow)
real(4),intent(in)::e
real(4),intent(in)::c
real(4),intent(in)::h
real(4),intent(out)::flow
end
end
