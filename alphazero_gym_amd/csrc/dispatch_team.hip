// persistent team kernel for wide networks, both environments
#include "team_dispatch.cuh"
hipError_t azg_team_dispatch_cartpole(azg_engine* e) { return team_dispatch<AZG_ENV_CARTPOLE>(e); }
hipError_t azg_team_dispatch_pendulum(azg_engine* e) { return team_dispatch<AZG_ENV_PENDULUM_V1>(e); }
