// persistent search kernels, CartPole (discrete MCTS)
#include "dispatch.cuh"
hipError_t azg_dispatch_cartpole(azg_engine* e) {
    hipError_t rc = dispatch_small<AZG_ENV_CARTPOLE>(e);
    return rc == hipErrorInvalidValue ? dispatch_large<AZG_ENV_CARTPOLE>(e) : rc;
}
