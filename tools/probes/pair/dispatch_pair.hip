// walker + server kernel pair for batches with more 16-tree groups than CUs, both environments
#include "pair_dispatch.cuh"
hipError_t azg_pair_dispatch_cartpole(azg_engine* e) { return pair_dispatch<AZG_ENV_CARTPOLE>(e); }
hipError_t azg_pair_dispatch_pendulum(azg_engine* e) { return pair_dispatch<AZG_ENV_PENDULUM_V1>(e); }
