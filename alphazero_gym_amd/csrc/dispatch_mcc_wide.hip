// lock-step path and persistent team kernel for wide networks, MountainCarContinuous
#include "ls_dispatch.cuh"
#include "team_dispatch.cuh"
hipError_t azg_ls_dispatch_mcc(azg_engine* e) { return ls_dispatch<AZG_ENV_MOUNTAINCAR_CONT>(e); }
hipError_t azg_team_dispatch_mcc(azg_engine* e) { return team_dispatch<AZG_ENV_MOUNTAINCAR_CONT>(e); }
