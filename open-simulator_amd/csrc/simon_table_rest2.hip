// simon_table_rest2.hip -- generation 6 of simon::table_kernel for 65 .. 128 internal node classes (simon_table.hip: template parameter CN2
// -- two node classes per lane in rest_select).  The device / device-less split doubles the node shapes of a gpushare cluster, so 33 shapes
// need it.  A translation unit of its own: build() runs one hipcc process per unit.
#define SIMON_TABLE_REST2_TU 1
#include "simon_table.hip"
