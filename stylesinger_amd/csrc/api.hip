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

SsTuning g_ss_tuning = {0, nullptr, 1, 0, 0, 1, 1, 1, 1};

extern "C" int ss_set_tuning(const char* key, int value) {
  if (!key) {
    ss_set_error("ss_set_tuning: null key");
    return SS_ERR_ARG;
  }
  if (strcmp(key, "wave_prio") == 0 && value >= 0 && value <= 2) {
    g_ss_tuning.wave_prio = value;
    return SS_OK;
  }
  if ((strcmp(key, "res16") == 0 || strcmp(key, "skip16") == 0) && (value == 0 || value == 1 || value == 4 || value == 6 || value == 8)) {
    (key[0] == 'r' ? g_ss_tuning.res16 : g_ss_tuning.skip16) = value;
    return SS_OK;
  }
  if (strcmp(key, "gate256") == 0 && (value == 0 || value == 1)) {
    g_ss_tuning.gate256 = value;
    return SS_OK;
  }
  if (strcmp(key, "gate16_ks") == 0 && (value == 0 || value == 1)) {
    g_ss_tuning.gate16_ks = value;
    return SS_OK;
  }
  if (strcmp(key, "gate16") == 0 && value >= 0 && value <= 3) {
    g_ss_tuning.gate16 = value;
    return SS_OK;
  }
  if ((strcmp(key, "res_tile") == 0 || strcmp(key, "skip_tile") == 0) && value >= SS_TILE_AUTO && value <= SS_TILE_128x32) {
    (key[0] == 'r' ? g_ss_tuning.res_tile : g_ss_tuning.skip_tile) = value;
    return SS_OK;
  }
  ss_set_error("ss_set_tuning: unknown key/value %s=%d", key, value);
  return SS_ERR_ARG;
}
extern "C" int ss_get_tuning(const char* key) {
  if (key) {
    if (strcmp(key, "wave_prio") == 0) return g_ss_tuning.wave_prio;
    if (strcmp(key, "gate16") == 0) return g_ss_tuning.gate16;
    if (strcmp(key, "gate256") == 0) return g_ss_tuning.gate256;
    if (strcmp(key, "gate16_ks") == 0) return g_ss_tuning.gate16_ks;
    if (strcmp(key, "res16") == 0) return g_ss_tuning.res16;
    if (strcmp(key, "skip16") == 0) return g_ss_tuning.skip16;
    if (strcmp(key, "res_tile") == 0) return g_ss_tuning.res_tile;
    if (strcmp(key, "skip_tile") == 0) return g_ss_tuning.skip_tile;
  }
  ss_set_error("ss_get_tuning: unknown key %s", key ? key : "(null)");
  return SS_ERR_ARG;
}
extern "C" int ss_set_clock_probe(void* dev_u64x2) {
  g_ss_tuning.clock_probe = static_cast<unsigned long long*>(dev_u64x2);
  return SS_OK;
}
extern "C" int ss_abi_version(void) { return SS_ABI_VERSION; }

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
  return SS_OK;
}
