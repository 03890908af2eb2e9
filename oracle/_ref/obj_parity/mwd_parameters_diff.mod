﻿!mod$ v1 sum:b688d5c5e1bddd84
!need$ 86b12428149ac79a n md_constant
!need$ 82a26416841665dd n mwd_setup
!need$ b7e498e07543ba78 n mwd_mesh
!need$ 76c1396aa4cc0721 n mwd_input_data
module mwd_parameters_diff
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
use mwd_mesh,only:meshdt
use mwd_mesh,only:meshdt_initialise
use mwd_setup,only:optimize_setupdt
use mwd_setup,only:setupdt
use mwd_setup,only:optimize_setupdt_initialise
use mwd_setup,only:setupdt_initialise
use mwd_input_data,only:input_datadt
use mwd_input_data,only:input_datadt_initialise
type::parametersdt
real(4),allocatable::ci(:,:)
real(4),allocatable::cp(:,:)
real(4),allocatable::beta(:,:)
real(4),allocatable::cft(:,:)
real(4),allocatable::cst(:,:)
real(4),allocatable::alpha(:,:)
real(4),allocatable::exc(:,:)
real(4),allocatable::b(:,:)
real(4),allocatable::cusl1(:,:)
real(4),allocatable::cusl2(:,:)
real(4),allocatable::clsl(:,:)
real(4),allocatable::ks(:,:)
real(4),allocatable::ds(:,:)
real(4),allocatable::dsm(:,:)
real(4),allocatable::ws(:,:)
real(4),allocatable::lr(:,:)
end type
type::hyper_parametersdt
real(4),allocatable::ci(:,:)
real(4),allocatable::cp(:,:)
real(4),allocatable::beta(:,:)
real(4),allocatable::cft(:,:)
real(4),allocatable::cst(:,:)
real(4),allocatable::alpha(:,:)
real(4),allocatable::exc(:,:)
real(4),allocatable::b(:,:)
real(4),allocatable::cusl1(:,:)
real(4),allocatable::cusl2(:,:)
real(4),allocatable::clsl(:,:)
real(4),allocatable::ks(:,:)
real(4),allocatable::ds(:,:)
real(4),allocatable::dsm(:,:)
real(4),allocatable::ws(:,:)
real(4),allocatable::lr(:,:)
end type
contains
subroutine parametersdt_initialise(this,mesh)
type(parametersdt),intent(inout)::this
type(meshdt),intent(in)::mesh
end
subroutine hyper_parametersdt_initialise(this,setup)
type(hyper_parametersdt),intent(inout)::this
type(setupdt),intent(in)::setup
end
end
