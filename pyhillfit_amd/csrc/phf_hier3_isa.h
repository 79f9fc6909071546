// phf_hier3_isa.h — the gfx950 assembly build of the hierarchical Ne = 3 iteration (phf_hier3_isa.hip loads it, phf_hierarchical.hip
// dispatches to it).  Internal to the library.
#ifndef PHF_HIER3_ISA_H
#define PHF_HIER3_ISA_H

#include <hip/hip_runtime.h>

#include "generated/phf_hier3_isa_layout.h"

// the kernels' unsigned division n / d: magic = min(floor(2^32 / d), 2^32 - 1); the estimate floor(n magic / 2^32) is exact or one low,
// the kernel corrects it once (tools/gen_hier_isa_main.py: udiv)
inline uint32_t phf_isa_magic(uint32_t d) { const uint64_t m = (1ULL << 32) / (uint64_t)d; return m > 0xffffffffULL ? 0xffffffffu : (uint32_t)m; }

// which kernel of the code object (generated/phf_hier3_isa_layout.h: phf_isa_hier_kernels[]) runs pairs of n_expts experiments whose
// point shape has this PHF_HIER_SHAPE code (phf_hier_points.points_per_expt): its index once the embedded code object is loaded on the
// current device and holds it, -1 if there is none
int phf_hier_isa_find(int n_expts, int shape_code);
// doubles per lane of device-memory scratch the kernel for (n_expts, shape_code) keeps part of a resident wavefront's chain state in
// (a->scratch: 512 bytes x slots x the launch's wavefronts); 0 for none or no such kernel.  A property of the generated table: no device call.
int phf_hier_isa_scratch_slots(int n_expts, int shape_code);
// launch it: `a` complete except `consts` (filled here); (grid_waves + 3) / 4 workgroups of 256 threads — grid_waves =
// a->total_waves for a plain launch (a->queue == NULL), the chip's wavefront slots for a queued one
int phf_hier_isa_advance(int which, phf_hier3_isa_args* a, int grid_waves, hipStream_t stream);

// launch phf_hier_fused_advance (every body in one persistent grid): `a` complete except `consts`
int phf_hier_isa_fused_advance(phf_hier_fused_args* a, int grid_waves, hipStream_t stream);

// the same for phf_sl3_advance: the single-level model-2 iteration (no moments), plain or queued
bool phf_sl3_isa_available();
int phf_sl3_isa_advance(phf_sl3_isa_args* a, int grid_waves, hipStream_t stream);

#endif
