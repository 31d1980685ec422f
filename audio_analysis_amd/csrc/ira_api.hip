// ABI version + error strings for libira.so.
#include "ira_common.h"

extern "C" int32_t ira_abi_version(void) { return IRA_ABI_VERSION; }

extern "C" const char* ira_error_string(int32_t code) {
  if (code == IRA_OK) return "ok";
  if (code == IRA_E_NULL) return "a required pointer argument was NULL";
  if (code == IRA_E_SIZE) return "a size or shape argument is outside the supported range";
  if (code == IRA_E_UNSUPPORTED) return "unsupported option";
  if (code == IRA_E_IO) return "file could not be opened or read";
  if (code == IRA_E_FORMAT) return "not a RIFF/WAVE file";
  if (code <= IRA_E_HIP_BASE) return hipGetErrorString((hipError_t)(IRA_E_HIP_BASE - code));
  return "unknown libira error";
}
