﻿!mod$ v1 sum:76c1396aa4cc0721
!need$ b7e498e07543ba78 n mwd_mesh
!need$ 86b12428149ac79a n md_constant
!need$ 82a26416841665dd n mwd_setup
module mwd_input_data
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
use mwd_mesh,only:meshdt
use mwd_mesh,only:meshdt_initialise
type::input_datadt
real(4),allocatable::qobs(:,:)
real(4),allocatable::prcp(:,:,:)
real(4),allocatable::pet(:,:,:)
real(4),allocatable::descriptor(:,:,:)
real(4),allocatable::sparse_prcp(:,:)
real(4),allocatable::sparse_pet(:,:)
real(4),allocatable::mean_prcp(:,:)
real(4),allocatable::mean_pet(:,:)
end type
contains
subroutine input_datadt_initialise(this,setup,mesh)
type(input_datadt),intent(inout)::this
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
end
end
