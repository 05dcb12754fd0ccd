// simon_table_team.hip -- the team-mode instantiations of simon::table_kernel (simon_table.hip: template parameter NW):
// kTeamWaves waves per scenario for batches too small to fill the chip with one wave each.  A translation unit of its own so that
// it compiles next to simon_table.hip (build(): one hipcc process per unit) instead of doubling that unit's three minutes.
#define SIMON_TABLE_TEAM_TU 1
#include "simon_table.hip"
