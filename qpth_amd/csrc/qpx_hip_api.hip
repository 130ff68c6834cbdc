// qpx_hip_api.hip -- the C ABI of include/qpx.h (argument checks + dispatch); the kernels and
// launchers it calls are compiled in qpx_hip_kernels.hip.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/qpx.h"
#include "qpx_launch.h"

#include "qpx_api.inc"
