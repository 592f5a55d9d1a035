#include "m3d_common.h"
#include "../../include/m3d_hip.h"
extern "C" int m3d_abi_version(void) { return M3D_ABI_VERSION; }
