﻿!mod$ v1 sum:ba196de66a1b48a6
!need$ c5f5068eb58aec21 n mwd_states
!need$ 82a26416841665dd n mwd_setup
!need$ 86b12428149ac79a n md_constant
!need$ 76c1396aa4cc0721 n mwd_input_data
module mwd_states_manipulation
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
use mwd_input_data,only:meshdt
use mwd_input_data,only:meshdt_initialise
use mwd_input_data,only:input_datadt
use mwd_input_data,only:input_datadt_initialise
use mwd_states,only:statesdt
use mwd_states,only:hyper_statesdt
use mwd_states,only:statesdt_initialise
use mwd_states,only:hyper_statesdt_initialise
interface set_states
procedure::set0d_states
procedure::set1d_states
procedure::set3d_states
end interface
interface set_hyper_states
procedure::set0d_hyper_states
procedure::set1d_hyper_states
procedure::set3d_hyper_states
end interface
contains
subroutine get_states(mesh,states,a)
type(meshdt),intent(in)::mesh
type(statesdt),intent(in)::states
real(4),intent(inout)::a(1_8:int(mesh%nrow,kind=8),1_8:int(mesh%ncol,kind=8),1_8:8_8)
end
subroutine set3d_states(mesh,states,a)
type(meshdt),intent(in)::mesh
type(statesdt),intent(inout)::states
real(4),intent(in)::a(1_8:int(mesh%nrow,kind=8),1_8:int(mesh%ncol,kind=8),1_8:8_8)
end
subroutine set1d_states(mesh,states,a)
type(meshdt),intent(in)::mesh
type(statesdt),intent(inout)::states
real(4),intent(in)::a(1_8:8_8)
end
subroutine set0d_states(mesh,states,a)
type(meshdt),intent(in)::mesh
type(statesdt),intent(inout)::states
real(4),intent(in)::a
end
subroutine normalize_states(setup,mesh,states)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(statesdt),intent(inout)::states
end
subroutine denormalize_states(setup,mesh,states)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(statesdt),intent(inout)::states
end
subroutine get_hyper_states(setup,hyper_states,a)
type(setupdt),intent(in)::setup
type(hyper_statesdt),intent(in)::hyper_states
real(4),intent(inout)::a(1_8:int(setup%optimize%nhyper,kind=8),1_8:1_8,1_8:8_8)
end
subroutine set3d_hyper_states(setup,hyper_states,a)
type(setupdt),intent(in)::setup
type(hyper_statesdt),intent(inout)::hyper_states
real(4),intent(in)::a(1_8:int(setup%optimize%nhyper,kind=8),1_8:1_8,1_8:8_8)
end
subroutine set1d_hyper_states(setup,hyper_states,a)
type(setupdt),intent(in)::setup
type(hyper_statesdt),intent(inout)::hyper_states
real(4),intent(in)::a(1_8:8_8)
end
subroutine set0d_hyper_states(setup,hyper_states,a)
type(setupdt),intent(in)::setup
type(hyper_statesdt),intent(inout)::hyper_states
real(4),intent(in)::a
end
subroutine hyper_states_to_states(hyper_states,states,setup,mesh,input_data)
type(hyper_statesdt),intent(in)::hyper_states
type(statesdt),intent(inout)::states
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
end
end
