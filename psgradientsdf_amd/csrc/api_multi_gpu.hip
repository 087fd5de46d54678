// api_multi_gpu.hip -- C ABI of the z-slab multi-GPU path: attaching a context to a rank and its communicator.
// The exchanges themselves are issued by the engine (comm.hip, loop.hip, engine.hip); the reference has no counterpart (single process).
#include "engine_internal.h"

using namespace psge;

extern "C" {
int psgsdf_comm_unique_id(uint8_t id[128]) { return id ? comm_unique_id(id) : PSGSDF_ERR_ARG; }
static int attach(psgsdf_ctx* c, int rank, int n_ranks) {
    if (!c || rank < 0 || n_ranks < 1 || rank >= n_ranks) return PSGSDF_ERR_ARG;
    c->rank = rank; c->n_ranks = n_ranks; c->inited = false; c->have_volume = false;      // (the slab is chosen when the volume is uploaded)
    return PSGSDF_OK;
}
int psgsdf_comm_init(psgsdf_ctx* c, const uint8_t id[128], int rank, int n_ranks) {
    int rc = attach(c, rank, n_ranks); if (rc) return rc;
    if (!id) return fail(c, PSGSDF_ERR_ARG, "comm_init: the id of psgsdf_comm_unique_id is required");
    return comm_create_rccl(c, id, rank, n_ranks);
}
int psgsdf_comm_init_ext(psgsdf_ctx* c, const psgsdf_comm_ops* ops, int rank, int n_ranks) {
    int rc = attach(c, rank, n_ranks); if (rc) return rc;
    return comm_create_ext(c, ops, rank, n_ranks);
}
int psgsdf_set_stream(psgsdf_ctx* c, void* hip_stream) {
    if (!c) return PSGSDF_ERR_ARG;
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->stream && c->own_stream) hipStreamDestroy(c->stream);
    c->stream = (hipStream_t)hip_stream; c->own_stream = false;
    return PSGSDF_OK;
}
int psgsdf_mg_info(psgsdf_ctx* c, int32_t out[12]) {
    if (!c || !(c->inited || c->have_volume)) return PSGSDF_ERR_STATE;
    if (!c->inited) { for (int i = 0; i < 6; ++i) out[i] = 0; out[5] = c->F; }      // (a volume without a band yet: the planes only)
    else { out[0] = (int32_t)c->S_global; out[1] = c->band.S; out[2] = c->row0; out[3] = c->row1; out[4] = c->halo; out[5] = c->F; } out[6] = c->rank; out[7] = c->n_ranks;
    out[8] = c->need[0]; out[9] = c->need[1]; out[10] = c->z0; out[11] = c->z1;
    return PSGSDF_OK;
}
int psgsdf_comm_stats(psgsdf_ctx* c, int64_t* n_collectives) { if (!c || !n_collectives) return PSGSDF_ERR_ARG; *n_collectives = c->n_collectives; return PSGSDF_OK; }
}  // extern "C"
