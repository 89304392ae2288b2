// Measurement aids of libcruller_hip.so (include/crl.h "measurement"): nothing on the product path launches these.
//
// crl_debug_occupy_cus: n workgroups that each claim ALL 160 KiB of a CU's LDS -- so no workgroup of any other kernel fits beside one --
// and sleep until a deadline or a stop flag.  On a single GPU this stands in for what RCCL's all-reduce kernels do to the training step
// of a data-parallel run (they hold CUs for as long as a gradient bucket is in flight): bench.py --occupy-cus N times the step with N
// CUs taken away, static against dynamic tile scheduling in the persistent GEMMs (profiles/README.md).
#include "common.h"

namespace {
__global__ __launch_bounds__(64) void occupy_kernel(unsigned long long ticks, const int* stop) {
  extern __shared__ char lds[];
  if (threadIdx.x == 0) {
    lds[0] = 1;                                                   // the allocation is real
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz, independent of the shader clock
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
      if (stop && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) break;
      __builtin_amdgcn_s_sleep(127);
    }
  }
}
}  // namespace

extern "C" int crl_debug_occupy_cus(int n_cus, double max_seconds, const int* stop_flag, void* stream) {
  CRL_CHECK(n_cus >= 1 && n_cus <= 255, "crl_debug_occupy_cus: n_cus %d outside [1, 255]", n_cus);
  CRL_CHECK(max_seconds > 0.0 && max_seconds <= 120.0, "crl_debug_occupy_cus: max_seconds must be in (0, 120]");
  if (int rc = crl_enable_lds(reinterpret_cast<const void*>(&occupy_kernel), 163840, "crl_debug_occupy_cus")) return rc;
  occupy_kernel<<<n_cus, 64, 163840, as_stream(stream)>>>((unsigned long long)(max_seconds * 1e8), stop_flag);
  CRL_LAUNCH_CHECK("crl_debug_occupy_cus");
  return 0;
}
