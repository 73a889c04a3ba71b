// mz_learn_conv_host.h -- what learner.hip (the C ABI of include/mzlearner.h) calls for net_kind == MZL_NET_BOARD: the conv-net learner of
// learner_conv.hip over the kernels of mz_learn_conv.h.  Internal to libmzlearner_hip.so.
#pragma once
#include <stdint.h>

#include <string>

#include "../../include/mzlearner.h"

struct mzlc_learner;

int mzlc_create(const mzl_config* cfg, int device_id, int num_cus, mzlc_learner** out, std::string& err);
void mzlc_destroy(mzlc_learner* h);
int64_t mzlc_num_params(const mzlc_learner* h);
int mzlc_num_tensors(const mzlc_learner* h);
int mzlc_tensor_info(const mzlc_learner* h, int i, const char** name, int64_t* offset, int32_t* rows, int32_t* cols);
int mzlc_num_buffers(const mzlc_learner* h);
int mzlc_buffer_info(const mzlc_learner* h, int i, const char** name, int64_t* offset, int32_t* count);
int64_t mzlc_num_running(const mzlc_learner* h);
int mzlc_bind(mzlc_learner* h, float* params, float* grads, float* m, float* v);
int mzlc_bind_buffers(mzlc_learner* h, float* running, int64_t* num_batches);
int mzlc_commit(mzlc_learner* h, void* stream, std::string& err);
int mzlc_grad(mzlc_learner* h, const mzl_batch* b, void* stream, std::string& err);
int mzlc_apply(mzlc_learner* h, double lr, double beta1, double beta2, double eps, double weight_decay, double max_grad_norm, int64_t step, void* stream,
               std::string& err);
int mzlc_debug_tensor(const mzlc_learner* h, const char* what, int a, int b, void** ptr, int64_t* count);
