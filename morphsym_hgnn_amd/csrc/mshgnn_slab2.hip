// Engine-driven stack kernels of the bf16 plan (mshgnn_wide_impl.hpp): slab2 geometry (16-window tiles, two workgroups per CU) and its entry points.
#define WD_PART 2
#include "mshgnn_wide_impl.hpp"
