// api.hip -- library-level entry points of libmvp_hip.so.
#include "common.h"

MVP_API const char* mvp_version(void) { return "mvp_hip 0.1 (gfx950, ROCm HIP, fp-contract=off)"; }

MVP_API const char* mvp_strerror(int code) {
  switch (code) {
    case MVP_OK: return "ok";
    case MVP_EINVAL: return "invalid argument (shape/size precondition violated)";
    case MVP_EUNSUPPORTED: return "valid request not supported by this build";
    case MVP_ENULL: return "required pointer is NULL";
  }
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "unknown mvp error";
}
