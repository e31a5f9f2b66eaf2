// lock-step path for wide networks, both environments
#include "ls_dispatch.cuh"
hipError_t azg_ls_dispatch_cartpole(azg_engine* e) { return ls_dispatch<AZG_ENV_CARTPOLE>(e); }
hipError_t azg_ls_dispatch_pendulum(azg_engine* e) { return ls_dispatch<AZG_ENV_PENDULUM_V1>(e); }
