// persistent search kernels, Acrobot-v1 (discrete MCTS, three actions, six observations), all hidden widths
#include "dispatch.cuh"
hipError_t azg_dispatch_acrobot(azg_engine* e) {
    hipError_t rc = dispatch_small<AZG_ENV_ACROBOT>(e);
    return rc == hipErrorInvalidValue ? dispatch_large<AZG_ENV_ACROBOT>(e) : rc;
}
