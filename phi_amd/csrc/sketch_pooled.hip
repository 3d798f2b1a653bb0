// sketch_pooled.hip -- the POOLED instances of the read sketch kernel (sketch.hip compiled a second time, for them alone:
// see the note at phi_launch_sketch_pooled there).  phi_amd/build.py compiles this file with -mllvm -disable-machine-licm.
#define PHI_SKETCH_POOLED_TU 1
#include "sketch.hip"
