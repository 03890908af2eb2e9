﻿!mod$ v1 sum:07043c518bcfdffd
!need$ b7e498e07543ba78 n mwd_mesh
!need$ 86b12428149ac79a n md_constant
module mw_mask
use md_constant,only:sp
use md_constant,only:gnp
use md_constant,only:gns
use mwd_mesh,only:meshdt
contains
recursive subroutine mask_upstream_cells(row,col,mesh,mask)
integer(4),intent(in)::row
integer(4),intent(in)::col
type(meshdt),intent(in)::mesh
logical(4),intent(inout)::mask(1_8:int(mesh%nrow,kind=8),1_8:int(mesh%ncol,kind=8))
end
end
