// simon_table_rs.hip -- generation 7 of simon::table_kernel over generation 6's position-mask rows (simon_table.hip: REST && SPREAD in one
// instantiation): soft PodTopologySpread constraints next to Open-Gpu-Share requests / required anti-affinity / ports / extra resources that
// do not fold into the table -- BASELINE config 5 as it is drawn, behind Services.  A translation unit of its own (one hipcc process per unit).
#define SIMON_TABLE_RS_TU 1
#include "simon_table.hip"
