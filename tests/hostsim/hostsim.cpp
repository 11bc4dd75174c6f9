// thread-local state of the host-side HIP simulator (tests only; see include/hip/hip_runtime.h)
#include <hip/hip_runtime.h>
namespace hostsim {
thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
thread_local BlockCtx* t_ctx = nullptr;
thread_local unsigned t_tid = 0;
}  // namespace hostsim

// The spectrogram encoder (mst_cnn*.hip: bf16 MFMA kernels) is not built into the simulator library; its C-ABI entry points
// exist so that the binding (which insists on every symbol of include/diffmst_hip.h) loads, and refuse to run.
#include "../../include/diffmst_hip.h"
extern "C" {
size_t mst_spectrogram_tables_bytes(void) { return 0; }
int mst_spectrogram_init_tables(void*, void*) { return 801; /* hipErrorNotSupported */ }
int mst_spectrogram_forward(const float*, int32_t, int64_t, int32_t, int32_t, const void*, float*, void*) { return 801; }
size_t mst_cnn14_workspace_bytes(const mst_cnn14_desc*) { return 0; }
int mst_cnn14_forward(const mst_cnn14_desc*, const float*, const mst_cnn14_params*, float*, float*, void*, size_t, void*) { return 801; }
int mst_cnn14_backward(const mst_cnn14_desc*, const float*, const mst_cnn14_params*, const float*, const mst_cnn14_grads*, void*, size_t, void*) { return 801; }
int mst_cnn14_forward_sync(const mst_cnn14_desc*, const float*, const mst_cnn14_params*, float*, float*, void*, size_t, void*, mst_sync_fn, void*) { return 801; }
int mst_cnn14_backward_sync(const mst_cnn14_desc*, const float*, const mst_cnn14_params*, const float*, const mst_cnn14_grads*, void*, size_t, void*, mst_sync_fn, void*) { return 801; }
// the controller's encoder stack (fp32 matrix-core kernels) is GPU-only as well
size_t mst_ctrl_workspace_bytes(const mst_ctrl_desc*) { return 0; }
int mst_ctrl_forward(const mst_ctrl_desc*, const float*, const uint8_t*, const mst_ctrl_layer*, float*, void*, size_t, void*) { return 801; }
int mst_ctrl_backward(const mst_ctrl_desc*, const float*, const mst_ctrl_layer*, const float*, const mst_ctrl_layer_grads*, float*, void*, size_t, void*) { return 801; }
int mst_ctrl_tokens_forward(const mst_ctrl_desc*, int32_t, const float*, const float*, const uint8_t*, const mst_ctrl_io*, float*, uint8_t*, void*) { return 801; }
int mst_ctrl_heads_forward(const mst_ctrl_desc*, int32_t, const float*, const mst_ctrl_io*, int32_t, int32_t, int32_t, float*, float*, float*, void*) { return 801; }
size_t mst_ctrl_heads_scratch_bytes(const mst_ctrl_desc*, int32_t) { return 0; }
int mst_ctrl_heads_backward(const mst_ctrl_desc*, int32_t, const float*, const mst_ctrl_io*, int32_t, int32_t, int32_t, const float*, const float*, const float*, const float*, const float*, const float*, const mst_ctrl_io_grads*, float*, void*, void*) { return 801; }
int mst_ctrl_tokens_backward(const mst_ctrl_desc*, int32_t, const float*, const mst_ctrl_io_grads*, void*) { return 801; }
}
