// simon_table_spread.hip -- generation 7 of simon::table_kernel: the single-wave instantiations with SPREAD (soft PodTopologySpread
// constraints, self-referential preferred pod (anti-)affinity, hard zone verdicts; simon_table.hip: template parameter SPREAD).  A
// translation unit of its own so that it compiles next to simon_table.hip (build(): one hipcc process per unit): together they were
// seven minutes of one process.
#define SIMON_TABLE_SPREAD_TU 1
#include "simon_table.hip"
