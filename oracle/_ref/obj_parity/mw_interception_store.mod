﻿!mod$ v1 sum:cf186a531d37cc55
!need$ d0630ccd2ecb08b1 n m_array_creation
!need$ eda5fd194b829f52 n mwd_parameters
!need$ 830568814217b360 n md_gr_operator
!need$ b74288d896965ed5 n mw_sparse_storage
!need$ 82a26416841665dd n mwd_setup
!need$ 76c1396aa4cc0721 n mwd_input_data
!need$ 86b12428149ac79a n md_constant
!need$ b7e498e07543ba78 n mwd_mesh
module mw_interception_store
use md_constant,only:sp
use m_array_creation,only:arange
use m_array_creation,only:linspace
use mwd_setup,only:setupdt
use mwd_mesh,only:meshdt
use mwd_input_data,only:input_datadt
use mwd_parameters,only:parametersdt
use md_gr_operator,only:gr_interception
use mw_sparse_storage,only:sparse_vector_to_matrix_r
use m_array_creation,only:m_array_creation$m_array_creation$arange_r=>arange_r
contains
subroutine adjust_interception_store(setup,mesh,input_data,parameters,nday,day_index)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
type(parametersdt),intent(inout)::parameters
integer(4),intent(in)::nday
integer(4),intent(in)::day_index(1_8:int(setup%ntime_step,kind=8))
end
end
