// simon_table_spread2.hip -- generation 7 of simon::table_kernel for 65 .. 128 internal node classes (simon_table.hip: template parameter
// CN2 -- two node classes per lane in spread_select).  The zone split multiplies the node shapes by the zones of the spread constraints'
// keys, so 25 shapes in 3 zones need it.  A translation unit of its own: build() runs one hipcc process per unit.
#define SIMON_TABLE_SPREAD2_TU 1
#include "simon_table.hip"
