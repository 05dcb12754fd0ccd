// The EXPLAIN form of the all-feature kernel's full variant (simon_wide.hip: wide_kernel<T, true, 1>; serves simon_explain) as a translation
// unit of its own: build() runs one hipcc process per unit.
#define SIMON_WIDE_EXPLAIN_TU 1
#include "simon_wide.hip"
