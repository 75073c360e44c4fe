// mtfjsp_encoder.hip — rollout forward passes of the job actor (GIN encoder + candidate scorer + local critic)
// and the machine actor (3x shared 2-node GAT + BatchNorm + scorer + local critic) for MI355X (gfx950).
// WORK IN PROGRESS in this commit: handle + weight management are real, the forward kernels land next.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <map>
#include <string>
#include <vector>

#include "../../include/mtfjsp.h"

struct mtfjsp_encoder {
    mtfjsp_encoder_config_t cfg;
    hipStream_t stream = nullptr;
    std::string err;
    std::map<std::string, float *> w;
    std::map<std::string, int64_t> wn;
    std::vector<void *> owned;
};
static thread_local std::string g_enc_err;

extern "C" const char *mtfjsp_encoder_last_error(mtfjsp_encoder_t e) { return e ? e->err.c_str() : g_enc_err.c_str(); }
extern "C" int mtfjsp_encoder_create(const mtfjsp_encoder_config_t *cfg, mtfjsp_encoder_t *out)
{
    if (!cfg || !out) { g_enc_err = "null argument"; return MTFJSP_ERR_ARG; }
    mtfjsp_encoder *e = new mtfjsp_encoder();
    e->cfg = *cfg;
    *out = e;
    return MTFJSP_OK;
}
extern "C" int mtfjsp_encoder_destroy(mtfjsp_encoder_t e)
{
    if (!e) return MTFJSP_OK;
    for (void *p : e->owned) (void)hipFree(p);
    delete e;
    return MTFJSP_OK;
}
extern "C" int mtfjsp_encoder_set_stream(mtfjsp_encoder_t e, void *s) { if (!e) return MTFJSP_ERR_ARG; e->stream = (hipStream_t)s; return MTFJSP_OK; }
extern "C" int mtfjsp_encoder_load_weight_host(mtfjsp_encoder_t e, const char *, const float *, int64_t) { if (!e) return MTFJSP_ERR_ARG; e->err = "encoder kernels not built yet"; return MTFJSP_ERR_STATE; }
extern "C" int mtfjsp_encoder_weights_ready(mtfjsp_encoder_t e) { if (!e) return MTFJSP_ERR_ARG; return MTFJSP_ERR_STATE; }
extern "C" int mtfjsp_job_actor_forward(mtfjsp_encoder_t e, const void *, const int32_t *, const float *, const int32_t *, const uint8_t *,
                                        const float *, float *, float *, float *, float *) { if (!e) return MTFJSP_ERR_ARG; e->err = "encoder kernels not built yet"; return MTFJSP_ERR_STATE; }
extern "C" int mtfjsp_machine_actor_forward(mtfjsp_encoder_t e, const void *, const void *, const float *, const uint8_t *, float *, float *, float *) { if (!e) return MTFJSP_ERR_ARG; e->err = "encoder kernels not built yet"; return MTFJSP_ERR_STATE; }
extern "C" int mtfjsp_sample_categorical(mtfjsp_encoder_t e, const float *, int32_t, int32_t, uint64_t, uint64_t, int32_t *, float *, const int32_t *, int32_t *) { if (!e) return MTFJSP_ERR_ARG; e->err = "encoder kernels not built yet"; return MTFJSP_ERR_STATE; }
extern "C" int mtfjsp_encoder_timing_begin(mtfjsp_encoder_t e) { return e ? MTFJSP_OK : MTFJSP_ERR_ARG; }
extern "C" int mtfjsp_encoder_timing_end(mtfjsp_encoder_t e, double *ms, int64_t *n) { if (!e) return MTFJSP_ERR_ARG; if (ms) *ms = 0; if (n) *n = 0; return MTFJSP_OK; }
