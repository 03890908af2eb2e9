﻿!mod$ v1 sum:82a26416841665dd
!need$ 86b12428149ac79a n md_constant
module mwd_setup
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
type::optimize_setupdt
character(128_4,1)::algorithm="...                                                                                                                             "
character(20_4,1),allocatable::jobs_fun(:)
real(4),allocatable::wjobs_fun(:)
real(4)::wjreg=0._4
character(20_4,1),allocatable::jreg_fun(:)
real(4),allocatable::wjreg_fun(:)
integer(4),allocatable::reg_descriptors_for_params(:,:)
integer(4),allocatable::reg_descriptors_for_states(:,:)
integer(4)::njf=0_4
integer(4)::njr=0_4
logical(4)::verbose=.true._4
character(128_4,1)::mapping="...                                                                                                                             "
logical(4)::denormalize_forward=.false._4
integer(4)::nhyper=0_4
integer(4)::optimize_start_step=1_4
integer(4)::maxiter=100_4
integer(4)::optim_parameters(1_8:16_8)=[INTEGER(4)::0_4,0_4,0_4,0_4,0_4,0_4,0_4,0_4,0_4,0_4,0_4,0_4,0_4,0_4,0_4,0_4]
integer(4)::optim_states(1_8:8_8)=[INTEGER(4)::0_4,0_4,0_4,0_4,0_4,0_4,0_4,0_4]
real(4)::lb_parameters(1_8:16_8)=[REAL(4)::9.999999974752427078783512115478515625e-7_4,9.999999974752427078783512115478515625e-7_4,9.999999974752427078783512115478515625e-7_4,9.999999974752427078783512115478515625e-7_4,9.999999974752427078783512115478515625e-7_4,9.999999974752427078783512115478515625e-7_4,-5.e1_4,9.999999974752427078783512115478515625e-7_4,9.999999974752427078783512115478515625e-7_4,9.999999974752427078783512115478515625e-7_4,9.999999974752427078783512115478515625e-7_4,9.999999974752427078783512115478515625e-7_4,9.999999974752427078783512115478515625e-7_4,9.999999974752427078783512115478515625e-7_4,9.999999974752427078783512115478515625e-7_4,9.999999974752427078783512115478515625e-7_4]
real(4)::ub_parameters(1_8:16_8)=[REAL(4)::1.e2_4,1.e3_4,1.e3_4,1.e3_4,1.e4_4,9.99998986721038818359375e-1_4,5.e1_4,1.e1_4,2.e3_4,2.e3_4,2.e3_4,1.e4_4,9.99998986721038818359375e-1_4,3.e1_4,9.99998986721038818359375e-1_4,1.e3_4]
real(4)::lb_states(1_8:8_8)=[REAL(4)::9.999999974752427078783512115478515625e-7_4,9.999999974752427078783512115478515625e-7_4,9.999999974752427078783512115478515625e-7_4,9.999999974752427078783512115478515625e-7_4,9.999999974752427078783512115478515625e-7_4,9.999999974752427078783512115478515625e-7_4,9.999999974752427078783512115478515625e-7_4,9.999999974752427078783512115478515625e-7_4]
real(4)::ub_states(1_8:8_8)=[REAL(4)::9.99998986721038818359375e-1_4,9.99998986721038818359375e-1_4,9.99998986721038818359375e-1_4,9.99998986721038818359375e-1_4,9.99998986721038818359375e-1_4,9.99998986721038818359375e-1_4,9.99998986721038818359375e-1_4,1.e4_4]
real(4),allocatable::wgauge(:)
integer(4),allocatable::mask_event(:,:)
end type
type::setupdt
character(128_4,1)::structure="gr-a                                                                                                                            "
real(4)::dt=3.6e3_4
character(128_4,1)::start_time="...                                                                                                                             "
character(128_4,1)::end_time="...                                                                                                                             "
logical(4)::sparse_storage=.false._4
logical(4)::read_qobs=.false._4
character(128_4,1)::qobs_directory="...                                                                                                                             "
logical(4)::read_prcp=.false._4
character(128_4,1)::prcp_format="tif                                                                                                                             "
logical(4)::prcp_yyyymmdd_access=.false._4
real(4)::prcp_conversion_factor=1._4
character(128_4,1)::prcp_directory="...                                                                                                                             "
logical(4)::read_pet=.false._4
character(128_4,1)::pet_format="tif                                                                                                                             "
real(4)::pet_conversion_factor=1._4
character(128_4,1)::pet_directory="...                                                                                                                             "
logical(4)::daily_interannual_pet=.false._4
logical(4)::mean_forcing=.true._4
logical(4)::read_descriptor=.false._4
character(128_4,1)::descriptor_format="tif                                                                                                                             "
character(128_4,1)::descriptor_directory="...                                                                                                                             "
character(20_4,1),allocatable::descriptor_name(:)
logical(4)::save_qsim_domain=.false._4
logical(4)::save_net_prcp_domain=.false._4
type(optimize_setupdt)::optimize
integer(4)::ntime_step=0_4
integer(4)::nd=0_4
integer(4)::ncpu=1_4
character(10_4,1)::parameters_name(1_8:16_8)=[CHARACTER(KIND=1,LEN=10)::"ci        ","cp        ","beta      ","cft       ","cst       ","alpha     ","exc       ","b         ","cusl1     ","cusl2     ","clsl      ","ks        ","ds        ","dsm       ","ws        ","lr        "]
character(10_4,1)::states_name(1_8:8_8)=[CHARACTER(KIND=1,LEN=10)::"hi        ","hp        ","hft       ","hst       ","husl1     ","husl2     ","hlsl      ","hlr       "]
end type
contains
subroutine optimize_setupdt_initialise(this,ntime_step,nd,ng,mapping,njf,njr)
type(optimize_setupdt),intent(inout)::this
integer(4),intent(in)::ntime_step
integer(4),intent(in)::nd
integer(4),intent(in)::ng
character(*,1),intent(in)::mapping
integer(4),intent(in)::njf
integer(4),intent(in)::njr
end
subroutine setupdt_initialise(this,nd,ng)
type(setupdt),intent(inout)::this
integer(4),intent(in)::nd
integer(4),intent(in)::ng
end
end
