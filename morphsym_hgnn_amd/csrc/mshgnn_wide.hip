// Engine-driven stack kernels of the bf16 plan (mshgnn_wide_impl.hpp): wide geometry (32-window tiles), up to 18 nodes per window, and the wide entry points.
#define WD_PART 0
#include "mshgnn_wide_impl.hpp"
