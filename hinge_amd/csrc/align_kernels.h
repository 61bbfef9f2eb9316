// HIP kernels of the overlap trim / classify step shared by `hinge maximal` and `hinge layout`
// (ProcessAlignment = LOverlap::trim_overlap + AddTypesAsymmetric).  Filled in below.
#pragma once
#include <hip/hip_runtime.h>
