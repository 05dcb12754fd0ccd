// simon_table_cls4.hip -- generation 4 of simon::table_kernel for 129 .. 256 internal node classes (simon_table.hip: kCls4 -- the class terms' re-base
// walks further groups of 64 classes).  Clusters whose nodes come in more than 128 distinct (allocatable, score column) shapes -- 160 in bench.py's
// `config3_classes160` -- stay on the score table instead of generation 1.  A translation unit of its own: the re-base costs registers inside the scheduling
// cycle that the kernels for <= 128 classes must not pay; build() runs one hipcc process per unit.
#define SIMON_TABLE_CLS4_TU 1
#include "simon_table.hip"
