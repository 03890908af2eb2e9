﻿!mod$ v1 sum:c95024be842ba28c
!need$ 76c1396aa4cc0721 n mwd_input_data
!need$ b74288d896965ed5 n mw_sparse_storage
!need$ 07043c518bcfdffd n mw_mask
!need$ 449ebe81f4cf5566 n m_array_manipulation
!need$ 32ec551519ad85cf n m_statistic
!need$ 82a26416841665dd n mwd_setup
!need$ b7e498e07543ba78 n mwd_mesh
!need$ 86b12428149ac79a n md_constant
module mw_forcing_statistic
use md_constant,only:sp
use md_constant,only:dp
use mwd_setup,only:setupdt
use mwd_mesh,only:meshdt
use mwd_input_data,only:input_datadt
use mw_sparse_storage,only:sparse_vector_to_matrix_r
use mw_mask,only:mask_upstream_cells
use m_array_manipulation,only:ma_flatten
use m_statistic,only:quantile
use m_array_manipulation,only:m_array_manipulation$m_array_manipulation$ma_flatten2d_r=>ma_flatten2d_r
use m_statistic,only:m_statistic$m_statistic$quantile1d_r=>quantile1d_r
contains
subroutine compute_mean_forcing(setup,mesh,input_data)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(inout)::input_data
end
subroutine compute_prcp_indices(setup,mesh,input_data,prcp_indices)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
real(4)::prcp_indices(1_8:4_8,1_8:int(mesh%ng,kind=8),1_8:int(setup%ntime_step,kind=8))
end
end
