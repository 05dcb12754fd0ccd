// Generation 6 of the score-table kernel with its position-mask rows, row totals and the canonical index of every position in LDS
// (simon_table.hip: template LDSX) -- the shape a gpushare sweep of a few cluster sizes gets (the whole batch resident at once, rows +
// summaries within 159 KB per scenario).  A translation unit of its own: build() runs one hipcc process per unit.
#define SIMON_TABLE_RESTLDS_TU 1
#include "simon_table.hip"
