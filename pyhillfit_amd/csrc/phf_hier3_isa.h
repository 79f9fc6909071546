// phf_hier3_isa.h — the gfx950 assembly build of the hierarchical Ne = 3 iteration (phf_hier3_isa.hip loads it, phf_hierarchical.hip
// dispatches to it).  Internal to the library.
#ifndef PHF_HIER3_ISA_H
#define PHF_HIER3_ISA_H

#include <hip/hip_runtime.h>

#include "generated/phf_hier3_isa_layout.h"

// true once the embedded code object is loaded on the current device and holds the advance kernel
bool phf_hier3_isa_available();
// launch phf_hier3_advance: `a` complete except `consts` (filled here); (grid_waves + 3) / 4 workgroups of 256 threads — grid_waves =
// a->total_waves for a plain launch (a->queue == NULL), the chip's wavefront slots for a queued one
int phf_hier3_isa_advance(phf_hier3_isa_args* a, int grid_waves, hipStream_t stream);

#endif
