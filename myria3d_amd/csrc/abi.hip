#include "m3d_common.h"
#include "../../include/m3d_hip.h"
extern "C" int m3d_abi_version(void) { return M3D_ABI_VERSION; }

// id of the hipGraph capture the stream is part of (0 when it is not capturing): lets the host side tell events
// recorded inside the CURRENT capture (waitable, become graph edges) from events of an earlier capture or of eager
// execution (must not be waited on while capturing) — HipRandLANet.prefetch_geometry
extern "C" int m3d_stream_capture_id(void* stream, uint64_t* id_out) {
  if (!id_out) return M3D_ERR_INVALID;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  if (hipStreamGetCaptureInfo((hipStream_t)stream, &st, &id) != hipSuccess) return M3D_ERR_LAUNCH;
  *id_out = st == hipStreamCaptureStatusActive ? (uint64_t)id : 0;
  return M3D_OK;
}
