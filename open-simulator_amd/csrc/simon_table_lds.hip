// Generation 4 of the score-table kernel with the scenario's workspace (byte table + node state) in LDS instead of HBM (simon_table.hip:
// template LDSWS) -- the shape small batches of small problems get: at most one scenario per CU, table + state + summaries within 159 KB.
// A translation unit of its own: build() runs one hipcc process per unit.
#define SIMON_TABLE_LDS_TU 1
#include "simon_table.hip"
