// The Open-Local variant of the all-feature kernel (simon_wide.hip: wide_kernel<T, EXPLAIN, 2>, both EXPLAIN forms) as a translation unit of
// its own: build() runs one hipcc process per unit.
#define SIMON_WIDE_LOCAL_TU 1
#include "simon_wide.hip"
