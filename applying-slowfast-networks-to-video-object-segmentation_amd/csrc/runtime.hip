// runtime.hip -- error convention and device check of the C ABI (include/sfvos.h).
#include <string.h>

#include "common.h"

namespace sfvos {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return SFVOS_E_LAUNCH;
  }
  return SFVOS_OK;
}

int current_device() {
  int dev = -1;
  return hipGetDevice(&dev) == hipSuccess ? dev : -1;
}

int device_cu_count() {
  static int cus[LdsAttrOnce::MAX_DEV] = {};
  const int dev = current_device();
  if (dev < 0 || dev >= LdsAttrOnce::MAX_DEV) return 256;
  int c = __atomic_load_n(&cus[dev], __ATOMIC_ACQUIRE);
  if (c == 0) {
    hipDeviceProp_t prop;
    c = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    __atomic_store_n(&cus[dev], c, __ATOMIC_RELEASE);
  }
  return c;
}

}  // namespace sfvos

extern "C" int sfvos_version(void) { return 300; }

extern "C" int sfvos_abi_sizes(int* sizes, int n) {
  const int v[7] = {(int)sizeof(sfvos_conv_desc), (int)sizeof(sfvos_pyramid), (int)sizeof(sfvos_levels),
                    (int)sizeof(sfvos_mse_table), (int)sizeof(sfvos_bn_running), (int)sizeof(sfvos_pack_item),
                    (int)sizeof(sfvos_planar_level)};
  for (int i = 0; i < 7 && i < n && sizes; ++i) sizes[i] = v[i];
  return 7;
}

extern "C" const char* sfvos_last_error(void) { return sfvos::g_err; }

extern "C" int sfvos_check_device(void) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) {
    sfvos::set_error("no HIP device: %s", hipGetErrorString(e));
    return SFVOS_E_NODEV;
  }
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) {
    sfvos::set_error("hipGetDeviceProperties failed: %s", hipGetErrorString(e));
    return SFVOS_E_NODEV;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    sfvos::set_error("device %d is %s; libsfvos is built for gfx950 only", dev, prop.gcnArchName);
    return SFVOS_E_NODEV;
  }
  return SFVOS_OK;
}
