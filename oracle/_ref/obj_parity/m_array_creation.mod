﻿!mod$ v1 sum:d0630ccd2ecb08b1
!need$ 86b12428149ac79a n md_constant
module m_array_creation
use md_constant,only:sp
interface arange
procedure::arange_i
procedure::arange_r
end interface
interface linspace
procedure::linspace_i
procedure::linspace_r
end interface
contains
subroutine arange_i(stt,stp,step,res)
integer(4),intent(in)::stt
integer(4),intent(in)::stp
real(4),intent(in)::step
real(4),intent(inout)::res(1_8:int(ceiling(real(stp-stt,kind=4)/step),kind=8))
end
subroutine arange_r(stt,stp,step,res)
real(4),intent(in)::stt
real(4),intent(in)::stp
real(4),intent(in)::step
real(4),intent(inout)::res(1_8:int(ceiling((stp-stt)/step),kind=8))
end
subroutine linspace_i(stt,stp,n,res)
integer(4),intent(in)::stt
integer(4),intent(in)::stp
integer(4),intent(in)::n
real(4),intent(inout)::res(1_8:int(n,kind=8))
end
subroutine linspace_r(stt,stp,n,res)
real(4),intent(in)::stt
real(4),intent(in)::stp
integer(4),intent(in)::n
real(4),intent(inout)::res(1_8:int(n,kind=8))
end
end
