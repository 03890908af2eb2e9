﻿!mod$ v1 sum:3d7fc12a1465e543
!need$ 86b12428149ac79a n md_constant
!need$ 82a26416841665dd n mwd_setup
!need$ b7e498e07543ba78 n mwd_mesh
!need$ 23b6a22028cfb5af n mwd_states_diff
module mwd_output_diff
use mwd_states_diff,only:statesdt
use mwd_states_diff,only:hyper_statesdt
use mwd_states_diff,only:statesdt_initialise
use mwd_states_diff,only:hyper_statesdt_initialise
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
use mwd_states_diff,only:input_datadt
use mwd_states_diff,only:input_datadt_initialise
type::outputdt
real(4),allocatable::qsim(:,:)
real(4),allocatable::qsim_domain(:,:,:)
real(4),allocatable::sparse_qsim_domain(:,:)
real(4),allocatable::net_prcp_domain(:,:,:)
real(4),allocatable::sparse_net_prcp_domain(:,:)
real(4)::cost=0._4
real(4)::cost_jobs=0._4
real(4)::cost_jreg=0._4
real(4)::cost_jobs_initial=0._4
real(4)::cost_jreg_initial=0._4
type(statesdt)::fstates
end type
type::outputdt_diff
real(4),allocatable::qsim(:,:)
end type
contains
subroutine outputdt_initialise(this,setup,mesh)
type(outputdt),intent(inout)::this
type(setupdt),intent(inout)::setup
type(meshdt),intent(inout)::mesh
end
end
