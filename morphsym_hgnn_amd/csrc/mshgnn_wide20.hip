// Engine-driven stack kernels of the bf16 plan (mshgnn_wide_impl.hpp): wide geometry, 19-20 nodes per window (MiniCheetah-K4).
#define WD_PART 1
#include "mshgnn_wide_impl.hpp"
