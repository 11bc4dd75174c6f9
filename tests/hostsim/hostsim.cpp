// thread-local state of the host-side HIP simulator (tests only; see include/hip/hip_runtime.h)
#include <hip/hip_runtime.h>
namespace hostsim {
thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
thread_local BlockCtx* t_ctx = nullptr;
thread_local unsigned t_tid = 0;
}  // namespace hostsim
