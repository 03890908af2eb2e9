﻿!mod$ v1 sum:618fc63b1b451b64
!need$ 86b12428149ac79a n md_constant
module md_vic_operator
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
subroutine vic_infiltration(prcp,cusl1,cusl2,b,husl1,husl2,runoff)
real(4),intent(in)::prcp
real(4),intent(in)::cusl1
real(4),intent(in)::cusl2
real(4),intent(in)::b
real(4),intent(inout)::husl1
real(4),intent(inout)::husl2
real(4),intent(out)::runoff
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
subroutine vic_interflow(n,cusl2,husl2,qi)
real(4),intent(in)::n
real(4),intent(in)::cusl2
real(4),intent(inout)::husl2
real(4),intent(out)::qi
end
subroutine vic_baseflow(clsl,ds,dsm,ws,hlsl,qb)
real(4),intent(in)::clsl
real(4),intent(in)::ds
real(4),intent(in)::dsm
real(4),intent(in)::ws
real(4),intent(inout)::hlsl
real(4),intent(out)::qb
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
subroutine linear_evapotranspiration(e,c,h,flow)
real(4),intent(in)::e
real(4),intent(in)::c
real(4),intent(in)::h
real(4),intent(out)::flow
end
end
