﻿!mod$ v1 sum:5280eece6cf2cbbc
!need$ 86b12428149ac79a n md_constant
module m_sort
use md_constant,only:sp
interface insertionsort
procedure::insertionsort_i
procedure::insertionsort_r
end interface
interface quicksort
procedure::quicksort_i
procedure::quicksort_r
end interface
contains
subroutine insertionsort_i(a)
integer(4),intent(inout)::a(:)
end
subroutine insertionsort_r(a)
real(4),intent(inout)::a(:)
end
recursive subroutine quicksort_i(a)
integer(4),intent(inout)::a(:)
end
recursive subroutine quicksort_r(a)
real(4),intent(inout)::a(:)
end
end
