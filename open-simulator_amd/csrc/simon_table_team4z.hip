// simon_table_team4z.hip -- the half of simon_table_team4.hip's instantiations with NonZero == request (template parameter Z = true): the team
// unit was the longest pole of the build by far, so its two halves compile as two hipcc processes (build(): one per unit).
#define SIMON_TABLE_TEAM_TU 4
#define SIMON_TABLE_NZEQ_HALF 1
#include "simon_table.hip"
