// Generation 6 of the score-table kernel (simon_table.hip: template REST -- position masks for Open-Gpu-Share devices, required
// (anti-)affinity, host ports, ephemeral storage / extended resources) as a translation unit of its own: its 32 instantiations are the
// largest kernels of the single-wave family, and build() runs one hipcc process per unit.
#define SIMON_TABLE_REST_TU 1
#include "simon_table.hip"
