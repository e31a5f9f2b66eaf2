// persistent search kernels, Pendulum (continuous MCTS), hidden widths 256 and up
#include "dispatch.cuh"
hipError_t azg_dispatch_pendulum_large(azg_engine* e) { return dispatch_large<AZG_ENV_PENDULUM_V1>(e); }
