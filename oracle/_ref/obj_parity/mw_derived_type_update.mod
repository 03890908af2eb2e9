﻿!mod$ v1 sum:3ddc4d6e0ee97fe9
!need$ 82a26416841665dd n mwd_setup
!need$ 86b12428149ac79a n md_constant
module mw_derived_type_update
use md_constant,only:sp
use md_constant,only:dp
use md_constant,only:glb_parameters
use md_constant,only:gub_parameters
use md_constant,only:glb_states
use md_constant,only:gub_states
use mwd_setup,only:optimize_setupdt
contains
subroutine reset_optimize_setup(this)
type(optimize_setupdt),intent(inout)::this
end
subroutine update_optimize_setup_optimize_args(this,mapping,ntime_step,nd,ng,njf)
type(optimize_setupdt),intent(inout)::this
character(*,1),intent(in)::mapping
integer(4),intent(in)::ntime_step
integer(4),intent(in)::nd
integer(4),intent(in)::ng
integer(4),intent(in)::njf
end
subroutine update_optimize_setup_optimize_options(this,njr)
type(optimize_setupdt),intent(inout)::this
integer(4),intent(in)::njr
end
end
