﻿!mod$ v1 sum:4d57810507808050
!need$ eda5fd194b829f52 n mwd_parameters
!need$ 86b12428149ac79a n md_constant
!need$ 82a26416841665dd n mwd_setup
!need$ b7e498e07543ba78 n mwd_mesh
!need$ 76c1396aa4cc0721 n mwd_input_data
module mwd_parameters_manipulation
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
use mwd_input_data,only:input_datadt
use mwd_input_data,only:input_datadt_initialise
use mwd_parameters,only:parametersdt
use mwd_parameters,only:hyper_parametersdt
use mwd_parameters,only:parametersdt_initialise
use mwd_parameters,only:hyper_parametersdt_initialise
interface set_parameters
procedure::set0d_parameters
procedure::set1d_parameters
procedure::set3d_parameters
end interface
interface set_hyper_parameters
procedure::set0d_hyper_parameters
procedure::set1d_hyper_parameters
procedure::set3d_hyper_parameters
end interface
contains
subroutine get_parameters(mesh,parameters,a)
type(meshdt),intent(in)::mesh
type(parametersdt),intent(in)::parameters
real(4),intent(inout)::a(1_8:int(mesh%nrow,kind=8),1_8:int(mesh%ncol,kind=8),1_8:16_8)
end
subroutine set3d_parameters(mesh,parameters,a)
type(meshdt),intent(in)::mesh
type(parametersdt),intent(inout)::parameters
real(4),intent(in)::a(1_8:int(mesh%nrow,kind=8),1_8:int(mesh%ncol,kind=8),1_8:16_8)
end
subroutine set1d_parameters(mesh,parameters,a)
type(meshdt),intent(in)::mesh
type(parametersdt),intent(inout)::parameters
real(4),intent(in)::a(1_8:16_8)
end
subroutine set0d_parameters(mesh,parameters,a)
type(meshdt),intent(in)::mesh
type(parametersdt),intent(inout)::parameters
real(4),intent(in)::a
end
subroutine normalize_parameters(setup,mesh,parameters)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(parametersdt),intent(inout)::parameters
end
subroutine denormalize_parameters(setup,mesh,parameters)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(parametersdt),intent(inout)::parameters
end
subroutine get_hyper_parameters(setup,hyper_parameters,a)
type(setupdt),intent(in)::setup
type(hyper_parametersdt),intent(in)::hyper_parameters
real(4),intent(inout)::a(1_8:int(setup%optimize%nhyper,kind=8),1_8:1_8,1_8:16_8)
end
subroutine set3d_hyper_parameters(setup,hyper_parameters,a)
type(setupdt),intent(in)::setup
type(hyper_parametersdt),intent(inout)::hyper_parameters
real(4),intent(in)::a(1_8:int(setup%optimize%nhyper,kind=8),1_8:1_8,1_8:16_8)
end
subroutine set1d_hyper_parameters(setup,hyper_parameters,a)
type(setupdt),intent(in)::setup
type(hyper_parametersdt),intent(inout)::hyper_parameters
real(4),intent(in)::a(1_8:16_8)
end
subroutine set0d_hyper_parameters(setup,hyper_parameters,a)
type(setupdt),intent(in)::setup
type(hyper_parametersdt),intent(inout)::hyper_parameters
real(4),intent(in)::a
end
subroutine hyper_parameters_to_parameters(hyper_parameters,parameters,setup,mesh,input_data)
type(hyper_parametersdt),intent(in)::hyper_parameters
type(parametersdt),intent(inout)::parameters
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
end
end
