﻿!mod$ v1 sum:c5f5068eb58aec21
!need$ 76c1396aa4cc0721 n mwd_input_data
!need$ 86b12428149ac79a n md_constant
!need$ b7e498e07543ba78 n mwd_mesh
module mwd_states
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
use mwd_mesh,only:optimize_setupdt
use mwd_mesh,only:setupdt
use mwd_mesh,only:optimize_setupdt_initialise
use mwd_mesh,only:setupdt_initialise
use mwd_mesh,only:meshdt
use mwd_mesh,only:meshdt_initialise
use mwd_input_data,only:input_datadt
use mwd_input_data,only:input_datadt_initialise
type::statesdt
real(4),allocatable::hi(:,:)
real(4),allocatable::hp(:,:)
real(4),allocatable::hft(:,:)
real(4),allocatable::hst(:,:)
real(4),allocatable::husl1(:,:)
real(4),allocatable::husl2(:,:)
real(4),allocatable::hlsl(:,:)
real(4),allocatable::hlr(:,:)
end type
type::hyper_statesdt
real(4),allocatable::hi(:,:)
real(4),allocatable::hp(:,:)
real(4),allocatable::hft(:,:)
real(4),allocatable::hst(:,:)
real(4),allocatable::husl1(:,:)
real(4),allocatable::husl2(:,:)
real(4),allocatable::hlsl(:,:)
real(4),allocatable::hlr(:,:)
end type
contains
subroutine statesdt_initialise(this,mesh)
type(statesdt),intent(inout)::this
type(meshdt),intent(in)::mesh
end
subroutine hyper_statesdt_initialise(this,setup)
type(hyper_statesdt),intent(inout)::this
type(setupdt),intent(in)::setup
end
end
