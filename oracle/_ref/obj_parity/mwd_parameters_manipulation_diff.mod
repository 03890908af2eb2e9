﻿!mod$ v1 sum:31b450e492b4e7b2
!need$ 86b12428149ac79a n md_constant
!need$ 82a26416841665dd n mwd_setup
!need$ b7e498e07543ba78 n mwd_mesh
!need$ 76c1396aa4cc0721 n mwd_input_data
!need$ b688d5c5e1bddd84 n mwd_parameters_diff
module mwd_parameters_manipulation_diff
use mwd_parameters_diff,only:parametersdt
use mwd_parameters_diff,only:hyper_parametersdt
use mwd_parameters_diff,only:parametersdt_initialise
use mwd_parameters_diff,only:hyper_parametersdt_initialise
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
interface set_parameters
procedure::set0d_parameters
procedure::set1d_parameters
procedure::set3d_parameters
end interface
interface set_parameters_d
procedure::set3d_parameters_d
end interface
interface set_parameters_b
procedure::set3d_parameters_b
end interface
interface set_hyper_parameters
procedure::set0d_hyper_parameters
procedure::set1d_hyper_parameters
procedure::set3d_hyper_parameters
end interface
contains
subroutine get_parameters_d(mesh,parameters,parameters_d,a,a_d)
type(meshdt),intent(in)::mesh
type(parametersdt),intent(in)::parameters
type(parametersdt),intent(in)::parameters_d
real(4),intent(inout)::a(1_8:int(mesh%nrow,kind=8),1_8:int(mesh%ncol,kind=8),1_8:16_8)
real(4),intent(inout)::a_d(1_8:int(mesh%nrow,kind=8),1_8:int(mesh%ncol,kind=8),1_8:16_8)
end
subroutine get_parameters_b(mesh,parameters,parameters_b,a,a_b)
type(meshdt),intent(in)::mesh
type(parametersdt),intent(in)::parameters
type(parametersdt)::parameters_b
real(4),intent(inout)::a(1_8:int(mesh%nrow,kind=8),1_8:int(mesh%ncol,kind=8),1_8:16_8)
real(4),intent(inout)::a_b(1_8:int(mesh%nrow,kind=8),1_8:int(mesh%ncol,kind=8),1_8:16_8)
end
subroutine get_parameters(mesh,parameters,a)
type(meshdt),intent(in)::mesh
type(parametersdt),intent(in)::parameters
real(4),intent(inout)::a(1_8:int(mesh%nrow,kind=8),1_8:int(mesh%ncol,kind=8),1_8:16_8)
end
subroutine set3d_parameters_d(mesh,parameters,parameters_d,a,a_d)
type(meshdt),intent(in)::mesh
type(parametersdt),intent(inout)::parameters
type(parametersdt),intent(inout)::parameters_d
real(4),intent(in)::a(1_8:int(mesh%nrow,kind=8),1_8:int(mesh%ncol,kind=8),1_8:16_8)
real(4),intent(in)::a_d(1_8:int(mesh%nrow,kind=8),1_8:int(mesh%ncol,kind=8),1_8:16_8)
end
subroutine set3d_parameters_b(mesh,parameters,parameters_b,a,a_b)
type(meshdt),intent(in)::mesh
type(parametersdt),intent(inout)::parameters
type(parametersdt),intent(inout)::parameters_b
real(4),intent(in)::a(1_8:int(mesh%nrow,kind=8),1_8:int(mesh%ncol,kind=8),1_8:16_8)
real(4)::a_b(1_8:int(mesh%nrow,kind=8),1_8:int(mesh%ncol,kind=8),1_8:16_8)
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
subroutine normalize_parameters_d(setup,mesh,parameters,parameters_d)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(parametersdt),intent(inout)::parameters
type(parametersdt),intent(inout)::parameters_d
end
subroutine normalize_parameters_b(setup,mesh,parameters,parameters_b)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(parametersdt),intent(inout)::parameters
type(parametersdt),intent(inout)::parameters_b
end
subroutine normalize_parameters(setup,mesh,parameters)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(parametersdt),intent(inout)::parameters
end
subroutine denormalize_parameters_d(setup,mesh,parameters,parameters_d)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(parametersdt),intent(inout)::parameters
type(parametersdt),intent(inout)::parameters_d
end
subroutine denormalize_parameters_b(setup,mesh,parameters,parameters_b)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(parametersdt),intent(inout)::parameters
type(parametersdt),intent(inout)::parameters_b
end
subroutine denormalize_parameters(setup,mesh,parameters)
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(parametersdt),intent(inout)::parameters
end
subroutine get_hyper_parameters_d(setup,hyper_parameters,hyper_parameters_d,a,a_d)
type(setupdt),intent(in)::setup
type(hyper_parametersdt),intent(in)::hyper_parameters
type(hyper_parametersdt),intent(in)::hyper_parameters_d
real(4),intent(inout)::a(1_8:int(setup%optimize%nhyper,kind=8),1_8:1_8,1_8:16_8)
real(4),intent(inout)::a_d(1_8:int(setup%optimize%nhyper,kind=8),1_8:1_8,1_8:16_8)
end
subroutine get_hyper_parameters_b(setup,hyper_parameters,hyper_parameters_b,a,a_b)
type(setupdt),intent(in)::setup
type(hyper_parametersdt),intent(in)::hyper_parameters
type(hyper_parametersdt)::hyper_parameters_b
real(4),intent(inout)::a(1_8:int(setup%optimize%nhyper,kind=8),1_8:1_8,1_8:16_8)
real(4),intent(inout)::a_b(1_8:int(setup%optimize%nhyper,kind=8),1_8:1_8,1_8:16_8)
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
subroutine hyper_parameters_to_parameters_d(hyper_parameters,hyper_parameters_d,parameters,parameters_d,setup,mesh,input_data)
type(hyper_parametersdt),intent(in)::hyper_parameters
type(hyper_parametersdt),intent(in)::hyper_parameters_d
type(parametersdt),intent(inout)::parameters
type(parametersdt),intent(inout)::parameters_d
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
end
subroutine hyper_parameters_to_parameters_b(hyper_parameters,hyper_parameters_b,parameters,parameters_b,setup,mesh,input_data)
type(hyper_parametersdt),intent(in)::hyper_parameters
type(hyper_parametersdt)::hyper_parameters_b
type(parametersdt),intent(inout)::parameters
type(parametersdt),intent(inout)::parameters_b
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
end
subroutine hyper_parameters_to_parameters(hyper_parameters,parameters,setup,mesh,input_data)
type(hyper_parametersdt),intent(in)::hyper_parameters
type(parametersdt),intent(inout)::parameters
type(setupdt),intent(in)::setup
type(meshdt),intent(in)::mesh
type(input_datadt),intent(in)::input_data
end
end
