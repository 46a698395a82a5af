// model.cpp — Model C ABI (model.rs:56-191, worker.rs:78-106).  Encoder kernels: encoder_kernels.hip.
#include "common.h"

using namespace pcv;

extern "C" {

void pcv_model_desc_minilm_l6(pcv_model_desc* d) {
    if (!d) return;
    d->vocab_size = 30522;
    d->hidden = 384;
    d->layers = 6;
    d->heads = 12;
    d->intermediate = 1536;
    d->max_positions = 512;
    d->type_vocab = 2;
    d->layer_norm_eps = 1e-12f;
    d->pooling = PCV_POOL_MEAN;
    d->normalize = 1;
    d->dense_out = 0;
    d->dense_activation = PCV_ACT_IDENTITY;
    d->max_seq_length = 256;
    d->compute = PCV_COMPUTE_F32;
}

#define PCV_TODO(name)                                                        \
    return guarded([&] { PCV_FAIL(PCV_ERR_UNSUPPORTED, name ": encoder not built yet"); })

pcv_status pcv_model_create(pcv_ctx*, const pcv_model_desc*, const char*, uint64_t, pcv_model**) { PCV_TODO("pcv_model_create"); }
pcv_status pcv_model_destroy(pcv_model*) { return PCV_OK; }
pcv_status pcv_model_output_dim(pcv_model*, int*) { PCV_TODO("pcv_model_output_dim"); }
pcv_status pcv_model_set_tensor(pcv_model*, const char*, const float*, int64_t) { PCV_TODO("pcv_model_set_tensor"); }
pcv_status pcv_model_get_tensor(pcv_model*, const char*, float*, int64_t, int64_t*) { PCV_TODO("pcv_model_get_tensor"); }
pcv_status pcv_model_encode_tokens(pcv_model*, const int64_t*, const int64_t*, int, int, float*) { PCV_TODO("pcv_model_encode_tokens"); }
pcv_status pcv_model_encode_tokens_device(pcv_model*, const int64_t*, const int64_t*, int, int, void*, int) { PCV_TODO("pcv_model_encode_tokens_device"); }
pcv_status pcv_model_debug_hidden(pcv_model*, int, float*, int64_t) { PCV_TODO("pcv_model_debug_hidden"); }
pcv_status pcv_model_last_stats(pcv_model*, pcv_encode_stats*) { PCV_TODO("pcv_model_last_stats"); }

}  // extern "C"
