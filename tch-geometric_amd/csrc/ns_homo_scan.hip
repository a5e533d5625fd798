// neighbor_sampling_homogenous with a temporal filter and/or the weighted
// sampler: every candidate edge must be inspected (timestamps / weights), so
// the work shape is one wavefront per frontier vertex streaming its column.
#include "tg_device.h"
#include "tg_host.h"

int tg_ns_homo_filtered_launch(const tg_graph *, const int64_t *, int64_t, int64_t, const int64_t *, int32_t,
                               const tg_ns_config *, const tg_rng *, const tg_ns_out *, hipStream_t) {
    return tg::fail(TG_ERR_UNSUPPORTED, "tg_ns_homo_batched: filtered / weighted sampling not built yet");
}
