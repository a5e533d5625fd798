// Calibration kernel for the measurement harness: the chip's ceiling for
// independent random 8-byte gathers from a table much larger than the
// Infinity Cache -- the access pattern of `indices[edge_ptr]` in neighbor
// sampling.  Not part of the sampling path.
#include "tg_device.h"
#include "tg_host.h"

namespace tg {
__global__ void gather_probe_kernel(const int64_t *table, int64_t n_table, int64_t per_thread, uint64_t seed,
                                    int64_t *sink) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const CallKey ck = call_key(seed, 0, 0xBEu);
    int64_t acc = 0;
    for (int64_t i = 0; i < per_thread; i += 8) {
        // 8 independent gathers in flight per lane, 4 Philox blocks
        int64_t v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const Draw d = draw(ck, (uint64_t)t, (uint32_t)(i / 2 + j), 0u);
            v[2 * j] = table[bounded64(d.a(), (uint64_t)n_table)];
            v[2 * j + 1] = table[bounded64(d.b(), (uint64_t)n_table)];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc ^= v[j];
    }
    sink[t] = acc;
}
} // namespace tg

extern "C" TG_API int tg_probe_random_gather(const int64_t *table, int64_t n_table, int64_t n_threads,
                                             int64_t per_thread, uint64_t seed, int64_t *sink, void *stream) {
    TG_REQUIRE(table && sink && n_table > 0 && n_threads > 0 && n_threads % 256 == 0 && per_thread % 8 == 0,
               "tg_probe_random_gather: bad arguments");
    hipLaunchKernelGGL(tg::gather_probe_kernel, dim3((unsigned)(n_threads / 256)), dim3(256), 0, (hipStream_t)stream,
                       table, n_table, per_thread, seed, sink);
    TG_LAUNCH_CHECK();
    return TG_OK;
}
