// simon_table_rsz.hip -- the half of simon_table_rs.hip's instantiations (REST && SPREAD) with NonZero == request (template parameter Z = true),
// a hipcc process of its own as simon_table_team4z.hip is.
#define SIMON_TABLE_RS_TU 1
#define SIMON_TABLE_NZEQ_HALF 1
#include "simon_table.hip"
