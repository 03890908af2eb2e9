﻿!mod$ v1 sum:9a5a3c5e64eb3f78
!need$ eda5fd194b829f52 n mwd_parameters
!need$ 44a770df04028c8f n mwd_output
!need$ 76c1396aa4cc0721 n mwd_input_data
!need$ 86b12428149ac79a n md_constant
!need$ 82a26416841665dd n mwd_setup
!need$ b7e498e07543ba78 n mwd_mesh
!need$ c5f5068eb58aec21 n mwd_states
module mw_derived_type_copy
use md_constant,only:sp
use md_constant,only:dp
use mwd_setup,only:setupdt
use mwd_mesh,only:meshdt
use mwd_input_data,only:input_datadt
use mwd_parameters,only:parametersdt
use mwd_states,only:statesdt
use mwd_output,only:outputdt
contains
subroutine copy_setup(this,copy)
type(setupdt),intent(in)::this
type(setupdt),intent(out)::copy
end
subroutine copy_mesh(this,copy)
type(meshdt),intent(in)::this
type(meshdt),intent(out)::copy
end
subroutine copy_input_data(this,copy)
type(input_datadt),intent(in)::this
type(input_datadt),intent(out)::copy
end
subroutine copy_parameters(this,copy)
type(parametersdt),intent(in)::this
type(parametersdt),intent(out)::copy
end
subroutine copy_states(this,copy)
type(statesdt),intent(in)::this
type(statesdt),intent(out)::copy
end
subroutine copy_output(this,copy)
type(outputdt),intent(in)::this
type(outputdt),intent(out)::copy
end
end
