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
Options& options() {
  static Options o;
  return o;
}
}  // namespace bcp

extern "C" int bcp_version(void) { return BCP_ABI_VERSION; }

// name = one of the Options fields (common.h); value = decimal integer(s), comma separated for the array-valued options;
// an empty value restores the default.  Not thread-safe against concurrent launches: set options before the work starts.
extern "C" int bcp_set_option(const char* name, const char* value) {
  BCP_REQUIRE(name && value, "bcp_set_option: null argument");
  bcp::Options& o = bcp::options();
  const bcp::Options def;
  long long v[5] = {0, 0, 0, 0, 0};
  const int n = sscanf(value, "%lld,%lld,%lld,%lld,%lld", &v[0], &v[1], &v[2], &v[3], &v[4]);
  const bool reset = n <= 0;
#define BCP_OPT_INT(field) if (!strcmp(name, #field)) { o.field = reset ? def.field : (int)v[0]; return BCP_OK; }
  BCP_OPT_INT(conv3_p) BCP_OPT_INT(splitk) BCP_OPT_INT(conv3_b6_flat_sk) BCP_OPT_INT(conv3_b6_cfg2d) BCP_OPT_INT(conv3_b6_w22) BCP_OPT_INT(conv3_b6_pipe) BCP_OPT_INT(conv3_b6_cin16max) BCP_OPT_INT(res_pcu) BCP_OPT_INT(res_nt) BCP_OPT_INT(wgrad_nt)
  BCP_OPT_INT(tn_groups) BCP_OPT_INT(cc_tile) BCP_OPT_INT(pack_sections) BCP_OPT_INT(cc_select_blocks) BCP_OPT_INT(conv3_b6) BCP_OPT_INT(conv3_b6_levels) BCP_OPT_INT(conv3_b6_direct) BCP_OPT_INT(conv3_b6_flat) BCP_OPT_INT(conv3_b6_minvox) BCP_OPT_INT(wgrad_b6) BCP_OPT_INT(wgrad_b6_minvox) BCP_OPT_INT(wgrad_b6_levels) BCP_OPT_INT(wgrad_b6_slots) BCP_OPT_INT(cc_border_dedupe) BCP_OPT_INT(cc_count_tile) BCP_OPT_INT(cc_fuse_select) BCP_OPT_INT(gemm_walk) BCP_OPT_INT(gemm_pipe) BCP_OPT_INT(gemm_stat_r) BCP_OPT_INT(norm_apply_cap) BCP_OPT_INT(norm_apply_vec) BCP_OPT_INT(k2_stats) BCP_OPT_INT(k2_bwd_stats) BCP_OPT_INT(up_recompute) BCP_OPT_INT(wgrad_b6_deep) BCP_OPT_INT(wgrad_b6_deep_nt) BCP_OPT_INT(wgrad_b6_deep_slots) BCP_OPT_INT(wgrad_b6_deep_tile) BCP_OPT_INT(norm_slabs) BCP_OPT_INT(conv3_f16) BCP_OPT_INT(fuse_bwd_stats) BCP_OPT_INT(conv3_xcd) BCP_OPT_INT(mix_c1) BCP_OPT_INT(whatif) BCP_OPT_INT(norm_own) BCP_OPT_INT(wgrad_reduce_flat) BCP_OPT_INT(norm_fin_rows) BCP_OPT_INT(norm_fuse_fin)
#undef BCP_OPT_INT
  if (!strcmp(name, "conv3_sk_elems")) { o.conv3_sk_elems = reset ? def.conv3_sk_elems : v[0]; return BCP_OK; }
  if (!strcmp(name, "res_tile2d_vox")) { o.res_tile2d_vox = reset ? def.res_tile2d_vox : v[0]; return BCP_OK; }
  if (!strcmp(name, "conv3_cfg")) {
    BCP_REQUIRE(reset || n == 4, "bcp_set_option: conv3_cfg wants TD,TH,TW,NT");
    for (int i = 0; i < 4; ++i) o.conv3_cfg[i] = reset ? 0 : (int)v[i];
    return BCP_OK;
  }
  if (!strcmp(name, "wgrad_tile")) {
    BCP_REQUIRE(reset || n >= 3, "bcp_set_option: wgrad_tile wants TD,TH,TW[,min_voxels,max_voxels]");
    for (int i = 0; i < 5; ++i) o.wgrad_tile[i] = def.wgrad_tile[i];
    for (int i = 0; i < n && !reset; ++i) o.wgrad_tile[i] = v[i];
    return BCP_OK;
  }
  bcp::set_error("bcp_set_option: unknown option '%s'", name);
  return BCP_EINVAL;
}
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

// ---- HIP graphs: a recorded launch plan (bcp_amd/plan.py) replayed under stream capture becomes ONE graph launch per network pass.
// Relaxed capture mode: the host framework's allocator may run on other threads while the pass is being captured.
extern "C" int bcp_graph_begin_capture(void* stream) {
  BCP_REQUIRE(stream, "bcp_graph_begin_capture: the default (null) stream cannot be captured");
  const hipError_t e = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeRelaxed);
  if (e != hipSuccess) { bcp::set_error("hipStreamBeginCapture: %s", hipGetErrorString(e)); return BCP_ELAUNCH; }
  return BCP_OK;
}

extern "C" int bcp_graph_end_capture(void* stream, void** graph_exec) {
  BCP_REQUIRE(stream && graph_exec, "bcp_graph_end_capture: null");
  *graph_exec = nullptr;
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture((hipStream_t)stream, &g);
  if (e != hipSuccess || !g) { bcp::set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); (void)hipGetLastError(); return BCP_ELAUNCH; }
  hipGraphExec_t x = nullptr;
  e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
  hipGraphDestroy(g);
  if (e != hipSuccess || !x) { bcp::set_error("hipGraphInstantiate: %s", hipGetErrorString(e)); (void)hipGetLastError(); return BCP_ELAUNCH; }
  *graph_exec = (void*)x;
  return BCP_OK;
}

extern "C" int bcp_graph_launch(void* graph_exec, void* stream) {
  BCP_REQUIRE(graph_exec, "bcp_graph_launch: null");
  const hipError_t e = hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
  if (e != hipSuccess) { bcp::set_error("hipGraphLaunch: %s", hipGetErrorString(e)); return BCP_ELAUNCH; }
  return BCP_OK;
}

extern "C" int bcp_graph_destroy(void* graph_exec) {
  if (graph_exec) hipGraphExecDestroy((hipGraphExec_t)graph_exec);
  return BCP_OK;
}
