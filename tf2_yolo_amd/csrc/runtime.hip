// Error reporting and device probe for the C-ABI (include/yolo_hip.h).
#include "conv_args.hpp"
#include <cstdlib>
#include <cstring>

namespace yolo {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Run-time options. Index = OPT_* (conv_args.hpp); defaults come from the environment once, yolo_set_option
// overrides them (benchmarks switch kernel variants inside one process: rule "A/B in one process").
int g_opt[OPT_COUNT];
void* g_dbg_buf = nullptr;
size_t g_dbg_bytes = 0;
static void options_from_env() {
  for (int i = 0; i < OPT_COUNT; ++i) g_opt[i] = 0;
  const char* e = getenv("YOLO_CONV_WIN");       // 0 off, 1 automatic, 2 force 128x128 tiles, 4 force 256x128 tiles
  g_opt[OPT_CONV_WIN] = e ? atoi(e) : 1;
  // launches that would leave most of the chip idle (needs yolo_set_conv_workspace): 0 = one workgroup per tile,
  // 1 = split-K with a reduce kernel (window and per-tap kernels), -1 = stream-K form of the window kernel (tile
  // tickets, the last arriver combines), > 1 = stream-K with that many workgroups (benchmarks, tests)
  e = getenv("YOLO_CONV_SK");
  g_opt[OPT_CONV_SK] = e ? atoi(e) : 1;
  // 3x3 stride-1 window kernel on 2-D patches (conv_win.hip, GEO = 1): 0 off, 1 automatic, 2 wherever the shape allows
  e = getenv("YOLO_CONV_PATCH");
  g_opt[OPT_CONV_PATCH] = e ? atoi(e) : 1;
  // 3x3 stride-1 filter gradient with the input window in LDS (conv_wgrad_win.hip): 0 off, wherever the shape allows: 1 = on
  // v_mfma_f32_32x32x16_f16, 2 = that with fragment reads two tap-steps ahead, 3 = on v_mfma_f32_16x16x32_f16 (default 1: 3 is 4 % faster alone, equal in the step)
  e = getenv("YOLO_WGRAD_WIN");
  g_opt[OPT_WGRAD_WIN] = e ? atoi(e) : 1;
  // hard / DIoU NMS: 0 = pair tests as a bit matrix by the whole chip + per-class walk over the bits (classes of up to 8192
  // rows; larger ones walk), 1 = the greedy walk kernel for every class (tests: both must give the same rows)
  e = getenv("YOLO_NMS_WALK");
  g_opt[OPT_NMS_WALK] = e ? atoi(e) : 0;
  e = getenv("YOLO_EXP");   // experiment / A-B bits (conv_args.hpp: OPT_EXP)
  g_opt[OPT_EXP] = e ? atoi(e) : 0;
}
void init_options() {
  static bool done = false;
  if (done) return;
  done = true;
  options_from_env();
}

}  // namespace yolo

extern "C" int yolo_set_option(int key, int value) {
  yolo::init_options();
  if (key == -1) {   // back to the defaults (environment)
    yolo::options_from_env();
    return YOLO_OK;
  }
  if (key < 0 || key >= yolo::OPT_COUNT) {
    yolo::set_error("yolo_set_option: unknown key %d", key);
    return YOLO_ERR_INVALID_ARG;
  }
  yolo::g_opt[key] = value;
  return YOLO_OK;
}

// tickets (1 MB) + two accumulator slabs (up to 256 x 128 fp32) for each of up to 512 workgroups
extern "C" size_t yolo_conv_workspace_bytes(void) { return (size_t)(1 << 20) + (size_t)512 * 2 * 256 * 128 * 4; }

extern "C" int yolo_set_conv_workspace(void* p, size_t bytes, void* stream) {
  yolo::init_options();
  return yolo::set_conv_workspace(p, bytes, yolo::as_stream(stream));
}

extern "C" int yolo_set_debug_buffer(void* p, size_t bytes) {
  yolo::g_dbg_buf = p;
  yolo::g_dbg_bytes = bytes;
  return YOLO_OK;
}

extern "C" const char* yolo_last_error(void) { return yolo::g_err; }

extern "C" int yolo_abi_version(void) { return 5; }   // 5: round 6 added yolo_bn_act_bwd_reduce_fold_ld (the reduction finished by its own launch), option key 8 (timing experiments), yolo_maxpool2x2_bwd, yolo_bn_act_maxpool2x2_fwd, yolo_split_planes_concat_ex, yolo_conv2d_fwd_head_unit; 4: round 5 added yolo_conv2d_dgrad_planes_bnred, yolo_bn_act_bwd_sum_partials, yolo_bnred_slots_cap, option key 7, yolo_allreduce_bucket; 3: round 4 added yolo_bn_act_bwd_reduce_bound_ld / _apply_planes_ld (dout with a row pitch), option key 6; 2: round 3 added yolo_cal_iou, yolo_nms_select, yolo_adam_step_dev, yolo_bn_finalize_offset, the wgrad workspace, yolo_mfma_probe

// ---- gradient exchange: the thin RCCL wrapper of SURVEY.md section 8b ----
// For a host that OWNS an RCCL communicator (a C++ trainer, a binding that creates its communicators itself): in-place
// sum all-reduce of one contiguous fp32 slice of the flat gradient buffer on the caller's stream. The library does not link
// RCCL: ncclAllReduce is looked up in the process at the first call (the copy the host already loaded -- torch bundles its
// own -- or librccl.so from the loader path), so there is never a second RCCL in the process. The Python host of this
// repository keeps using torch.distributed (tf2_yolo_amd/dp.py): torch does not hand out its ncclComm_t.
#include <dlfcn.h>
namespace {
using nccl_allreduce_fn = int (*)(const void*, void*, size_t, int, int, void*, hipStream_t);
nccl_allreduce_fn find_nccl_allreduce() {
  void* f = dlsym(RTLD_DEFAULT, "ncclAllReduce");
  if (f == nullptr) {
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (h == nullptr) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (h != nullptr) f = dlsym(h, "ncclAllReduce");
  }
  return reinterpret_cast<nccl_allreduce_fn>(f);
}
}  // namespace

extern "C" int yolo_allreduce_bucket(void* rccl_comm, float* grads, long long count, void* stream) {
  YOLO_REQUIRE(rccl_comm != nullptr && grads != nullptr && count > 0, "allreduce_bucket: bad args");
  static nccl_allreduce_fn fn = find_nccl_allreduce();
  if (fn == nullptr) {
    yolo::set_error("allreduce_bucket: no RCCL in this process (ncclAllReduce not found, librccl.so not loadable)");
    return YOLO_ERR_LAUNCH;
  }
  constexpr int kNcclFloat = 7, kNcclSum = 0;   // rccl.h: ncclFloat32, ncclSum
  const int rc = fn(grads, grads, (size_t)count, kNcclFloat, kNcclSum, rccl_comm, yolo::as_stream(stream));
  if (rc != 0) {
    yolo::set_error("allreduce_bucket: ncclAllReduce returned %d", rc);
    return YOLO_ERR_LAUNCH;
  }
  return YOLO_OK;
}

extern "C" int yolo_device_available(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n > 0 ? 1 : 0;
}
