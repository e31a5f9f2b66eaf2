// batched network inference (azg_mlp_eval), all widths
#include "engine_host.h"
#include "mlp_eval.cuh"
hipError_t azg_dispatch_mlp_eval(azg_engine* e, const float* obs, int n, float* value, float* dist, float* raw) {
    switch (e->HP) {
        case 64: return mlp_eval_launch<64>(e, obs, n, value, dist, raw);
        case 128: return mlp_eval_launch<128>(e, obs, n, value, dist, raw);
        case 256: return mlp_eval_launch<256>(e, obs, n, value, dist, raw);
        case 512: return mlp_eval_launch<512>(e, obs, n, value, dist, raw);
        case 1024: return mlp_eval_launch<1024>(e, obs, n, value, dist, raw);
    }
    return hipErrorInvalidValue;
}
