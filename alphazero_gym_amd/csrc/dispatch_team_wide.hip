// persistent team kernel, config E's network (4x1024, Pendulum, LDS trees): the forms for batches beyond two 32-tree workgroups per CU
#define AZG_TEAM_WIDE_TU
#include "team_dispatch.cuh"
