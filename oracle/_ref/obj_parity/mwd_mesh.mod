﻿!mod$ v1 sum:b7e498e07543ba78
!need$ 82a26416841665dd n mwd_setup
!need$ 86b12428149ac79a n md_constant
module mwd_mesh
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
use mwd_setup,only:optimize_setupdt
use mwd_setup,only:setupdt
use mwd_setup,only:optimize_setupdt_initialise
use mwd_setup,only:setupdt_initialise
type::meshdt
real(4)::dx
integer(4)::nrow
integer(4)::ncol
integer(4)::ng
integer(4)::nac
integer(4)::xmin
integer(4)::ymax
integer(4),allocatable::flwdir(:,:)
integer(4),allocatable::flwacc(:,:)
integer(4),allocatable::path(:,:)
integer(4),allocatable::active_cell(:,:)
real(4),allocatable::flwdst(:,:)
integer(4),allocatable::gauge_pos(:,:)
character(20_4,1),allocatable::code(:)
real(4),allocatable::area(:)
integer(4),allocatable::rowcol_to_ind_sparse(:,:)
integer(4),allocatable::local_active_cell(:,:)
end type
contains
subroutine meshdt_initialise(this,setup,nrow,ncol,ng)
type(meshdt),intent(inout)::this
type(setupdt),intent(inout)::setup
integer(4),intent(in)::nrow
integer(4),intent(in)::ncol
integer(4),intent(in)::ng
end
end
