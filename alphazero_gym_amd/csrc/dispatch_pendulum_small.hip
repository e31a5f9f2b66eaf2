// persistent search kernels, Pendulum (continuous MCTS), hidden widths up to 128
#include "dispatch.cuh"
hipError_t azg_dispatch_pendulum_small(azg_engine* e) { return dispatch_small<AZG_ENV_PENDULUM_V1>(e); }
