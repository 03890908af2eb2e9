﻿!mod$ v1 sum:449ebe81f4cf5566
!need$ 86b12428149ac79a n md_constant
module m_array_manipulation
use md_constant,only:sp
interface ma_flatten
procedure::ma_flatten2d_i
procedure::ma_flatten2d_r
procedure::ma_flatten3d_i
procedure::ma_flatten3d_r
end interface
interface flatten
procedure::flatten2d_i
procedure::flatten2d_r
procedure::flatten3d_i
procedure::flatten3d_r
end interface
contains
subroutine ma_flatten2d_i(a,mask,res)
integer(4),intent(in)::a(:,:)
logical(4),intent(in)::mask(1_8:size(a,dim=1,kind=8),1_8:size(a,dim=2,kind=8))
integer(4),allocatable,intent(inout)::res(:)
end
subroutine ma_flatten2d_r(a,mask,res)
real(4),intent(in)::a(:,:)
logical(4),intent(in)::mask(1_8:size(a,dim=1,kind=8),1_8:size(a,dim=2,kind=8))
real(4),allocatable,intent(inout)::res(:)
end
subroutine ma_flatten3d_i(a,mask,res)
integer(4),intent(in)::a(:,:,:)
logical(4),intent(in)::mask(1_8:size(a,dim=1,kind=8),1_8:size(a,dim=2,kind=8),1_8:size(a,dim=3,kind=8))
integer(4),allocatable,intent(inout)::res(:)
end
subroutine ma_flatten3d_r(a,mask,res)
real(4),intent(in)::a(:,:,:)
logical(4),intent(in)::mask(1_8:size(a,dim=1,kind=8),1_8:size(a,dim=2,kind=8),1_8:size(a,dim=3,kind=8))
real(4),allocatable,intent(inout)::res(:)
end
subroutine flatten2d_i(a,res)
integer(4),intent(in)::a(:,:)
integer(4),intent(inout)::res(1_8:int(int(size(a,dim=1,kind=8)*size(a,dim=2,kind=8),kind=4),kind=8))
end
subroutine flatten2d_r(a,res)
real(4),intent(in)::a(:,:)
real(4),intent(inout)::res(1_8:int(int(size(a,dim=1,kind=8)*size(a,dim=2,kind=8),kind=4),kind=8))
end
subroutine flatten3d_i(a,res)
integer(4),intent(in)::a(:,:,:)
integer(4),intent(inout)::res(1_8:int(int(size(a,dim=1,kind=8)*size(a,dim=2,kind=8)*size(a,dim=3,kind=8),kind=4),kind=8))
end
subroutine flatten3d_r(a,res)
real(4),intent(in)::a(:,:,:)
real(4),intent(inout)::res(1_8:int(int(size(a,dim=1,kind=8)*size(a,dim=2,kind=8)*size(a,dim=3,kind=8),kind=4),kind=8))
end
end
