// Error reporting + device sanity for libstylesinger_hip.
#include "common.h"
#include "../../include/stylesinger_hip.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void ss_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* ss_last_error(void) { return g_err; }

SsTuning g_ss_tuning = {0, nullptr, 1, 0, 0, 1, 1, 1, 1, 0, 0, 0, 2048, 1, 1, 1, 0, 1, 1, 1};

namespace {
struct Knob { const char* key; int* slot; bool (*ok)(int); };
bool ok_01(int v) { return v == 0 || v == 1; }
bool ok_012(int v) { return v >= 0 && v <= 2; }
bool ok_0123(int v) { return v >= 0 && v <= 3; }
bool ok_mt(int v) { return v == 0 || v == 1 || v == 4 || v == 6 || v == 8; }
bool ok_tile(int v) { return v >= SS_TILE_AUTO && v <= SS_TILE_128x32; }
bool ok_htile(int v) { return v == 0 || v == 64 || v == 128; }
bool ok_mb(int v) { return v >= 1 && v <= 2048; }
const Knob* knobs(int* n) {
  static const Knob k[] = {
      {"wave_prio", &g_ss_tuning.wave_prio, ok_012},   {"gate16", &g_ss_tuning.gate16, ok_0123},     {"gate16_ks", &g_ss_tuning.gate16_ks, ok_01},
      {"gate256", &g_ss_tuning.gate256, ok_01},        {"res16", &g_ss_tuning.res16, ok_mt},          {"skip16", &g_ss_tuning.skip16, ok_mt},
      {"res_tile", &g_ss_tuning.res_tile, ok_tile},    {"skip_tile", &g_ss_tuning.skip_tile, ok_tile}, {"htile", &g_ss_tuning.htile, ok_htile},
      {"wino_tn", &g_ss_tuning.wino_tn, ok_012},       {"wino_v1", &g_ss_tuning.wino_v1, ok_01},      {"voc_wino_max_mb", &g_ss_tuning.voc_wino_max_mb, ok_mb},
      {"e16", &g_ss_tuning.e16, ok_01},                {"mel_tail", &g_ss_tuning.mel_tail, ok_01},     {"gate128", &g_ss_tuning.gate128, ok_01},
      {"q4_force", &g_ss_tuning.q4_force, ok_01},      {"layer512", &g_ss_tuning.layer512, ok_012},
      {"layer512_tail", &g_ss_tuning.layer512_tail, ok_012},       {"skip_dense", &g_ss_tuning.skip_dense, ok_01},
  };
  *n = (int)(sizeof(k) / sizeof(k[0]));
  return k;
}
}  // namespace

extern "C" int ss_set_tuning(const char* key, int value) {
  if (!key) {
    ss_set_error("ss_set_tuning: null key");
    return SS_ERR_ARG;
  }
  int n;
  const Knob* k = knobs(&n);
  for (int i = 0; i < n; ++i)
    if (strcmp(key, k[i].key) == 0 && k[i].ok(value)) {
      *k[i].slot = value;
      return SS_OK;
    }
  ss_set_error("ss_set_tuning: unknown key/value %s=%d", key, value);
  return SS_ERR_ARG;
}
extern "C" int ss_get_tuning(const char* key) {
  int n;
  const Knob* k = knobs(&n);
  for (int i = 0; key && i < n; ++i)
    if (strcmp(key, k[i].key) == 0) return *k[i].slot;
  ss_set_error("ss_get_tuning: unknown key %s", key ? key : "(null)");
  return SS_ERR_ARG;
}
extern "C" int ss_set_clock_probe(void* dev_u64x2) {
  g_ss_tuning.clock_probe = static_cast<unsigned long long*>(dev_u64x2);
  return SS_OK;
}
// measurement aid: an empty kernel with the launch geometry of a real one - the floor of a dependent launch edge (tools/launch_floor.py)
namespace {
__global__ void null_kernel(int* sink) {
  if (sink && blockIdx.x == 0x7fffffff) *sink = 0;   // never true: keeps the kernel from being optimised to nothing at all
}
}  // namespace
extern "C" int ss_debug_null_launch(int grid, int block, void* stream) {
  SS_CHECK_ARG(grid > 0 && block > 0 && block <= 1024, "ss_debug_null_launch: grid > 0, 0 < block <= 1024");
  hipLaunchKernelGGL(null_kernel, dim3(grid), dim3(block), 0, (hipStream_t)stream, (int*)nullptr);
  SS_CHECK_LAUNCH("ss_debug_null_launch");
  return SS_OK;
}

unsigned int* g_ss_q4_guard = nullptr;
extern "C" int ss_set_q4_guard(void* dev_u32x2) {
  g_ss_q4_guard = static_cast<unsigned int*>(dev_u32x2);
  return SS_OK;
}
extern "C" int ss_abi_version(void) { return SS_ABI_VERSION; }

int ss_n_cu() {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    hipDeviceProp_t prop;
    cached[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return cached[dev];
}

extern "C" int ss_device_info(int dev, int* n_cu, char* arch, int arch_len) {
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) {
    ss_set_error("ss_device_info: %s", hipGetErrorString(e));
    return SS_ERR_HIP;
  }
  if (n_cu) *n_cu = prop.multiProcessorCount;
  if (arch && arch_len > 0) {
    strncpy(arch, prop.gcnArchName, arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  return SS_OK;
}

// sizes of the public structs, so the ctypes mirror in stylesinger_amd/lib.py can be verified at load time
extern "C" int ss_struct_sizes(int64_t* out, int n) {
  if (!out || n < 3) {
    ss_set_error("ss_struct_sizes: need room for 3 entries");
    return SS_ERR_ARG;
  }
  out[0] = (int64_t)sizeof(ss_conv_gemm_args);
  out[1] = (int64_t)sizeof(ss_wavenet);
  out[2] = (int64_t)sizeof(ss_hifigan);
  if (n >= 4) out[3] = (int64_t)sizeof(ss_gemm_bf16_args);
  if (n >= 5) out[4] = (int64_t)sizeof(ss_f0track_params);
  if (n >= 6) out[5] = (int64_t)sizeof(ss_layer512_args);
  return SS_OK;
}
