﻿!mod$ v1 sum:32ec551519ad85cf
!need$ 5280eece6cf2cbbc n m_sort
!need$ 86b12428149ac79a n md_constant
module m_statistic
use md_constant,only:sp
use m_sort,only:quicksort
use m_sort,only:m_sort$m_sort$quicksort_i=>quicksort_i
use m_sort,only:m_sort$m_sort$quicksort_r=>quicksort_r
interface quantile
procedure::quantile0d_i
procedure::quantile0d_r
procedure::quantile1d_i
procedure::quantile1d_r
end interface
interface mean
procedure::mean1d_i
procedure::mean1d_r
procedure::mean2d_i
procedure::mean2d_r
end interface
interface variance
procedure::variance1d_i
procedure::variance1d_r
procedure::variance2d_i
procedure::variance2d_r
end interface
interface std
procedure::std1d_i
procedure::std1d_r
procedure::std2d_i
procedure::std2d_r
end interface
contains
subroutine quantile0d_i(a,q,res)
integer(4),intent(in)::a(:)
real(4),intent(in)::q
real(4),intent(inout)::res
end
subroutine quantile0d_r(a,q,res)
real(4),intent(in)::a(:)
real(4),intent(in)::q
real(4),intent(inout)::res
end
subroutine quantile1d_i(a,q,res)
integer(4),intent(in)::a(:)
real(4),intent(in)::q(:)
real(4),intent(inout)::res(1_8:size(q,dim=1,kind=8))
end
subroutine quantile1d_r(a,q,res)
real(4),intent(in)::a(:)
real(4),intent(in)::q(:)
real(4),intent(inout)::res(1_8:size(q,dim=1,kind=8))
end
subroutine mean1d_i(a,res)
integer(4),intent(in)::a(:)
real(4),intent(inout)::res
end
subroutine mean1d_r(a,res)
real(4),intent(in)::a(:)
real(4),intent(inout)::res
end
subroutine mean2d_i(a,res)
integer(4),intent(in)::a(:,:)
real(4),intent(inout)::res
end
subroutine mean2d_r(a,res)
real(4),intent(in)::a(:,:)
real(4),intent(inout)::res
end
subroutine variance1d_i(a,res)
integer(4),intent(in)::a(:)
real(4),intent(inout)::res
end
subroutine variance1d_r(a,res)
real(4),intent(in)::a(:)
real(4),intent(inout)::res
end
subroutine variance2d_i(a,res)
integer(4),intent(in)::a(:,:)
real(4),intent(inout)::res
end
subroutine variance2d_r(a,res)
real(4),intent(in)::a(:,:)
real(4),intent(inout)::res
end
subroutine std1d_i(a,res)
integer(4),intent(in)::a(:)
real(4),intent(inout)::res
end
subroutine std1d_r(a,res)
real(4),intent(in)::a(:)
real(4),intent(inout)::res
end
subroutine std2d_i(a,res)
integer(4),intent(in)::a(:,:)
real(4),intent(inout)::res
end
subroutine std2d_r(a,res)
real(4),intent(in)::a(:,:)
real(4),intent(inout)::res
end
end
