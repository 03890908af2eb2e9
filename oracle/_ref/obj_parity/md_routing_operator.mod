﻿!mod$ v1 sum:904794908a6ac3c0
!need$ 86b12428149ac79a n md_constant
module md_routing_operator
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
subroutine upstream_discharge(dt,dx,nrow,ncol,flwdir,flwacc,row,col,q,qup)
real(4),intent(in)::dt
real(4),intent(in)::dx
integer(4),intent(in)::nrow
integer(4),intent(in)::ncol
integer(4),intent(in)::flwdir(1_8:int(nrow,kind=8),1_8:int(ncol,kind=8))
integer(4),intent(in)::flwacc(1_8:int(nrow,kind=8),1_8:int(ncol,kind=8))
integer(4),intent(in)::row
integer(4),intent(in)::col
real(4),intent(in)::q(1_8:int(nrow,kind=8),1_8:int(ncol,kind=8))
real(4),intent(out)::qup
end
subroutine linear_routing(dt,qup,lr,hr,qrout)
real(4),intent(in)::dt
real(4),intent(in)::qup
real(4),intent(in)::lr
real(4),intent(inout)::hr
real(4),intent(out)::qrout
end
end
