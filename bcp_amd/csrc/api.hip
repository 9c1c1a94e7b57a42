// bcp_amd/csrc/api.hip -- library-level entry points of libbcp_hip.so (version, errors, device, events).
#include "common.h"
#include "../../include/bcp_hip.h"
#include <cstring>

namespace bcp {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace bcp

extern "C" int bcp_version(void) { return 100; }
extern "C" const char* bcp_last_error(void) { return bcp::g_err; }

extern "C" int bcp_device_arch(char* buf, int n) {
  BCP_REQUIRE(buf && n > 0, "bcp_device_arch: bad buffer");
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    bcp::set_error("bcp_device_arch: no HIP device");
    return BCP_ELAUNCH;
  }
  strncpy(buf, prop.gcnArchName, (size_t)n - 1);
  buf[n - 1] = 0;
  return BCP_OK;
}

extern "C" int bcp_event_create(void** ev) {
  BCP_REQUIRE(ev, "bcp_event_create: null");
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) { bcp::set_error("hipEventCreate failed"); return BCP_ELAUNCH; }
  *ev = (void*)e;
  return BCP_OK;
}
extern "C" int bcp_event_record(void* ev, void* stream) {
  if (hipEventRecord((hipEvent_t)ev, (hipStream_t)stream) != hipSuccess) { bcp::set_error("hipEventRecord failed"); return BCP_ELAUNCH; }
  return BCP_OK;
}
extern "C" int bcp_event_elapsed_ms(void* start, void* stop, float* ms) {
  BCP_REQUIRE(ms, "bcp_event_elapsed_ms: null");
  if (hipEventSynchronize((hipEvent_t)stop) != hipSuccess || hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop) != hipSuccess) {
    bcp::set_error("hipEventElapsedTime failed");
    return BCP_ELAUNCH;
  }
  return BCP_OK;
}
extern "C" int bcp_event_destroy(void* ev) {
  hipEventDestroy((hipEvent_t)ev);
  return BCP_OK;
}
