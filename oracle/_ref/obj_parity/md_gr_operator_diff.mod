﻿!mod$ v1 sum:54215331fdee0196
!need$ 86b12428149ac79a n md_constant
module md_gr_operator_diff
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
subroutine gr_interception_d(prcp,pet,ci,ci_d,hi,hi_d,pn,pn_d,ei,ei_d)
real(4),intent(in)::prcp
real(4),intent(in)::pet
real(4),intent(in)::ci
real(4),intent(in)::ci_d
real(4),intent(inout)::hi
real(4),intent(inout)::hi_d
real(4),intent(out)::pn
real(4),intent(out)::pn_d
real(4),intent(out)::ei
real(4),intent(out)::ei_d
end
subroutine gr_interception_b(prcp,pet,ci,ci_b,hi,hi_b,pn,pn_b,ei,ei_b)
real(4),intent(in)::prcp
real(4),intent(in)::pet
real(4),intent(in)::ci
real(4)::ci_b
real(4),intent(inout)::hi
real(4),intent(inout)::hi_b
real(4)::pn
real(4)::pn_b
real(4)::ei
real(4)::ei_b
end
subroutine gr_interception(prcp,pet,ci,hi,pn,ei)
real(4),intent(in)::prcp
real(4),intent(in)::pet
real(4),intent(in)::ci
real(4),intent(inout)::hi
real(4),intent(out)::pn
real(4),intent(out)::ei
end
subroutine gr_production_d(pn,pn_d,en,en_d,cp,cp_d,beta,hp,hp_d,pr,pr_d,perc,perc_d)
real(4),intent(in)::pn
real(4),intent(in)::pn_d
real(4),intent(in)::en
real(4),intent(in)::en_d
real(4),intent(in)::cp
real(4),intent(in)::cp_d
real(4),intent(in)::beta
real(4),intent(inout)::hp
real(4),intent(inout)::hp_d
real(4),intent(out)::pr
real(4),intent(out)::pr_d
real(4),intent(out)::perc
real(4),intent(out)::perc_d
end
subroutine gr_production_b(pn,pn_b,en,en_b,cp,cp_b,beta,hp,hp_b,pr,pr_b,perc,perc_b)
real(4),intent(in)::pn
real(4)::pn_b
real(4),intent(in)::en
real(4)::en_b
real(4),intent(in)::cp
real(4)::cp_b
real(4),intent(in)::beta
real(4),intent(inout)::hp
real(4),intent(inout)::hp_b
real(4)::pr
real(4)::pr_b
real(4)::perc
real(4)::perc_b
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
subroutine gr_exchange_d(exc,exc_d,hft,hft_d,l,l_d)
real(4),intent(in)::exc
real(4),intent(in)::exc_d
real(4),intent(inout)::hft
real(4),intent(inout)::hft_d
real(4),intent(out)::l
real(4),intent(out)::l_d
end
subroutine gr_exchange_b(exc,exc_b,hft,hft_b,l,l_b)
real(4),intent(in)::exc
real(4)::exc_b
real(4),intent(inout)::hft
real(4),intent(inout)::hft_b
real(4)::l
real(4)::l_b
end
subroutine gr_exchange(exc,hft,l)
real(4),intent(in)::exc
real(4),intent(inout)::hft
real(4),intent(out)::l
end
subroutine gr_transfer_d(n,prcp,pr,pr_d,ct,ct_d,ht,ht_d,q,q_d)
real(4),intent(in)::n
real(4),intent(in)::prcp
real(4),intent(in)::pr
real(4),intent(in)::pr_d
real(4),intent(in)::ct
real(4),intent(in)::ct_d
real(4),intent(inout)::ht
real(4),intent(inout)::ht_d
real(4),intent(out)::q
real(4),intent(out)::q_d
end
subroutine gr_transfer_b(n,prcp,pr,pr_b,ct,ct_b,ht,ht_b,q,q_b)
real(4),intent(in)::n
real(4),intent(in)::prcp
real(4),intent(in)::pr
real(4)::pr_b
real(4),intent(in)::ct
real(4)::ct_b
real(4),intent(inout)::ht
real(4),intent(inout)::ht_b
real(4)::q
real(4)::q_b
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
