﻿!mod$ v1 sum:830568814217b360
!need$ 86b12428149ac79a n md_constant
module md_gr_operator
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
subroutine gr_interception(prcp,pet,ci,hi,pn,ei)
real(4),intent(in)::prcp
real(4),intent(in)::pet
real(4),intent(in)::ci
real(4),intent(inout)::hi
real(4),intent(out)::pn
real(4),intent(out)::ei
end
subroutine gr_production(pn,en,cp,beta,hp,pr,perc)
real(4),intent(in)::pn
real(4),intent(in)::en
real(4),intent(in)::cp
real(4),intent(in)::beta
real(4),intent(inout)::hp
real(4),intent(out)::pr
real(4),intent(out)::perc
end
subroutine gr_exchange(exc,hft,l)
real(4),intent(in)::exc
real(4),intent(inout)::hft
real(4),intent(out)::l
end
subroutine gr_transfer(n,prcp,pr,ct,ht,q)
real(4),intent(in)::n
real(4),intent(in)::prcp
real(4),intent(in)::pr
real(4),intent(in)::ct
real(4),intent(inout)::ht
real(4),intent(out)::q
end
end
