﻿!mod$ v1 sum:b74288d896965ed5
!need$ b7e498e07543ba78 n mwd_mesh
!need$ 86b12428149ac79a n md_constant
module mw_sparse_storage
use md_constant,only:sp
use md_constant,only:dp
use mwd_mesh,only:meshdt
contains
subroutine compute_rowcol_to_ind_sparse(mesh)
type(meshdt),intent(inout)::mesh
end
subroutine sparse_matrix_to_vector_r(mesh,matrix,vector)
type(meshdt),intent(in)::mesh
real(4),intent(in)::matrix(1_8:int(mesh%nrow,kind=8),1_8:int(mesh%ncol,kind=8))
real(4),intent(inout)::vector(1_8:int(mesh%nac,kind=8))
end
subroutine sparse_matrix_to_vector_i(mesh,matrix,vector)
type(meshdt),intent(in)::mesh
integer(4),intent(in)::matrix(1_8:int(mesh%nrow,kind=8),1_8:int(mesh%ncol,kind=8))
integer(4),intent(inout)::vector(1_8:int(mesh%nac,kind=8))
end
subroutine sparse_vector_to_matrix_r(mesh,vector,matrix,na_value)
type(meshdt),intent(in)::mesh
real(4),intent(in)::vector(1_8:int(mesh%nac,kind=8))
real(4),intent(inout)::matrix(1_8:int(mesh%nrow,kind=8),1_8:int(mesh%ncol,kind=8))
real(4),intent(in),optional::na_value
end
subroutine sparse_vector_to_matrix_i(mesh,vector,matrix,na_value)
type(meshdt),intent(in)::mesh
integer(4),intent(in)::vector(1_8:int(mesh%nac,kind=8))
integer(4),intent(inout)::matrix(1_8:int(mesh%nrow,kind=8),1_8:int(mesh%ncol,kind=8))
integer(4),intent(in),optional::na_value
end
end
