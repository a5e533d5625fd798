#include "tg_device.h"
#include "tg_host.h"

extern "C" int tg_random_walk(const tg_graph *, const int64_t *, int64_t, int64_t, float, float, const tg_rng *,
                              int64_t *, void *) {
    return tg::fail(TG_ERR_UNSUPPORTED, "tg_random_walk: not built yet");
}
extern "C" int tg_tempo_random_walk(const tg_graph *, const int64_t *, const int64_t *, const int64_t *,
                                    const int64_t *, int64_t, int64_t, int64_t, int64_t, const tg_rng *, int64_t *,
                                    int64_t *, void *) {
    return tg::fail(TG_ERR_UNSUPPORTED, "tg_tempo_random_walk: not built yet");
}
