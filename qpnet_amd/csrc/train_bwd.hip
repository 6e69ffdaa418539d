// train_bwd.hip -- backward of the teacher-forced forward (placeholder until the kernels land)
#include "train_common.h"
#include "qpn_handle.h"

int qpn_launch_bwd(const TrainParams& p, const TrainBwd& bw, hipStream_t stream) {
    qpn_set_error("backward not built yet");
    return QPN_EINVAL;
}
extern "C" int qpn_train_backward(qpn_handle* h, const float* d_dlogits, float* d_flatgrad, void* stream) {
    qpn_set_error("backward not built yet");
    return QPN_EINVAL;
}
extern "C" int qpn_adam_step(qpn_handle* h, float* d_flat, const float* d_grad, float* d_m, float* d_v, int64_t n,
                             int step, float lr, float beta1, float beta2, float eps, float weight_decay, void* stream) {
    qpn_set_error("adam not built yet");
    return QPN_EINVAL;
}
