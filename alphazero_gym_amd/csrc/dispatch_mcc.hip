// persistent search kernels, MountainCarContinuous (continuous MCTS whose traces can end in terminal nodes), all hidden widths
#include "dispatch.cuh"
hipError_t azg_dispatch_mcc(azg_engine* e) {
    hipError_t rc = dispatch_small<AZG_ENV_MOUNTAINCAR_CONT>(e);
    return rc == hipErrorInvalidValue ? dispatch_large<AZG_ENV_MOUNTAINCAR_CONT>(e) : rc;
}
