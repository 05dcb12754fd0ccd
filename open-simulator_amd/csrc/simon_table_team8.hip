// simon_table_team8.hip -- the team-mode instantiations of simon::table_kernel with NW = 8 waves per scenario (simon_table.hip:
// template parameter NW) for batches too small to fill the chip with one wave each.  A translation unit of its own so that it
// compiles next to simon_table.hip (build(): one hipcc process per unit) instead of adding to that unit's three minutes.
#define SIMON_TABLE_TEAM_TU 8
#include "simon_table.hip"
