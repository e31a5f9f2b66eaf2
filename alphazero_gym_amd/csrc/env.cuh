// env.cuh -- closed-form CartPole / MountainCar / Pendulum dynamics on the device (float64; operation order = alphazero_gym_amd/envs.py).
#pragma once
#include "records.h"

// ------------------------------------------------------------------------------------------------ environments

// The kernels are instantiated per env FAMILY (template parameter ENV):
//   AZG_ENV_CARTPOLE          discrete actions (CartPole, MountainCar-v0: the step is chosen at run time by KParams::env_id)
//   AZG_ENV_PENDULUM_V1       one continuous action, never terminal (both Pendulum versions: KParams::v1)
//   AZG_ENV_MOUNTAINCAR_CONT  one continuous action, episodes END (MountainCarContinuous-v0): the continuous descent has a terminal
//                             exit and a trace may end in an existing terminal node (mcts.py:619-623, 682) -- compiled into this
//                             family only, so that the Pendulum kernels carry no exit mask
//   AZG_ENV_ACROBOT           discrete actions, SIX observations (Acrobot-v1): the network's first layer takes a second MFMA k-step and
//                             the workgroup's input block has eight rows; a family of its own so that the CartPole family's kernels
//                             carry neither (nor the Runge-Kutta dynamics)
template <int ENV> struct EnvFamily {
    static constexpr bool CONT = ENV == AZG_ENV_PENDULUM_V1 || ENV == AZG_ENV_MOUNTAINCAR_CONT;   // MCTSContinuous (progressive widening) vs MCTSDiscrete
    static constexpr bool TERM = ENV != AZG_ENV_PENDULUM_V1;         // nodes can be terminal
    static constexpr int S = CONT ? 2 : 4;                           // env state slots the kernels carry
    static constexpr bool IN8 = ENV == AZG_ENV_ACROBOT;              // five to eight network inputs
};

// observation of a state; Pendulum also returns sin(theta) so that the node can cache it for its children's dynamics
template <int ENV>
__device__ __forceinline__ void env_obs(const double* s, float* obs, double* sn_out) {
    if (ENV == AZG_ENV_CARTPOLE) {
        obs[0] = (float)s[0]; obs[1] = (float)s[1]; obs[2] = (float)s[2]; obs[3] = (float)s[3];
        *sn_out = 0.0;
    } else if (ENV == AZG_ENV_MOUNTAINCAR_CONT) {
        obs[0] = (float)s[0]; obs[1] = (float)s[1]; obs[2] = 0.0f; obs[3] = 0.0f;   // (position, velocity)
        *sn_out = 0.0;
    } else {
        double sn, cs;
        azg_sincos(s[0], &sn, &cs);
        obs[0] = (float)cs; obs[1] = (float)sn; obs[2] = (float)s[1]; obs[3] = 0.0f;
        *sn_out = sn;
    }
}

// gym CartPoleEnv.step (explicit Euler); same operation order as oracle/azg_oracle.c cartpole_step
__device__ __forceinline__ void cartpole_step(const double* s, int action, double* o, double* reward, int* done) {
    const double gravity = 9.8, masspole = 0.1, total_mass = 0.1 + 1.0, length = 0.5;
    const double polemass_length = 0.1 * 0.5, force_mag = 10.0, tau = 0.02;
    const double theta_thr = 12.0 * 2.0 * 3.141592653589793 / 360.0, x_thr = 2.4;
    double x = s[0], x_dot = s[1], theta = s[2], theta_dot = s[3];
    double force = action == 1 ? force_mag : -force_mag;
    double sintheta, costheta;
    azg_sincos(theta, &sintheta, &costheta);
    double temp = (force + (polemass_length * (theta_dot * theta_dot)) * sintheta) / total_mass;
    double thetaacc = (gravity * sintheta - costheta * temp) /
                      (length * (4.0 / 3.0 - (masspole * (costheta * costheta)) / total_mass));
    double xacc = temp - ((polemass_length * thetaacc) * costheta) / total_mass;
    x = x + tau * x_dot;
    x_dot = x_dot + tau * xacc;
    theta = theta + tau * theta_dot;
    theta_dot = theta_dot + tau * thetaacc;
    o[0] = x; o[1] = x_dot; o[2] = theta; o[3] = theta_dot;
    *done = (x < -x_thr) || (x > x_thr) || (theta < -theta_thr) || (theta > theta_thr);
    *reward = 1.0;
}

// gym MountainCarEnv.step (MountainCar-v0; three actions); same operation order as oracle/azg_oracle.c mountaincar_step.
// The state uses slots 0..1 of the discrete family's 4-slot state vector (slots 2..3 stay zero, and so do observations 2..3).
__device__ __forceinline__ void mountaincar_step(const double* s, int action, double* o, double* reward, int* done) {
    const double min_position = -1.2, max_position = 0.6, max_speed = 0.07, goal_position = 0.5, goal_velocity = 0.0;
    const double force = 0.001, gravity = 0.0025;
    double position = s[0], velocity = s[1];
    double sn, cs;
    azg_sincos(3.0 * position, &sn, &cs);
    velocity = velocity + ((double)(action - 1) * force + cs * (-gravity));
    velocity = velocity < -max_speed ? -max_speed : (velocity > max_speed ? max_speed : velocity);
    position = position + velocity;
    position = position < min_position ? min_position : (position > max_position ? max_position : position);
    if (position == min_position && velocity < 0.0) velocity = 0.0;
    o[0] = position; o[1] = velocity; o[2] = 0.0; o[3] = 0.0;
    *done = (position >= goal_position) && (velocity >= goal_velocity);
    *reward = -1.0;
}

// gym Continuous_MountainCarEnv.step (MountainCarContinuous-v0); same operation order as oracle/azg_oracle.c mountaincar_cont_step
// and alphazero_gym_amd/envs.py MountainCarContinuousEnv.step.  The force is the action clipped to [-1, 1]; the reward's action cost
// uses the action as it came (gym: math.pow(action[0], 2) * 0.1); +100 on reaching the flag.
__device__ __forceinline__ void mountaincar_cont_step(const double* s, float action, double* o, double* reward, int* done) {
    const double min_position = -1.2, max_position = 0.6, max_speed = 0.07, goal_position = 0.45, goal_velocity = 0.0, power = 0.0015;
    double position = s[0], velocity = s[1];
    const double a = (double)action;
    const double force = a < -1.0 ? -1.0 : (a > 1.0 ? 1.0 : a);
    double sn, cs;
    azg_sincos(3.0 * position, &sn, &cs);
    velocity = velocity + (force * power - 0.0025 * cs);
    velocity = velocity > max_speed ? max_speed : velocity;
    velocity = velocity < -max_speed ? -max_speed : velocity;
    position = position + velocity;
    position = position > max_position ? max_position : position;
    position = position < min_position ? min_position : position;
    if (position == min_position && velocity < 0.0) velocity = 0.0;
    const int d = (position >= goal_position) && (velocity >= goal_velocity);
    o[0] = position; o[1] = velocity;
    *done = d;
    *reward = (d ? 100.0 : 0.0) - (a * a) * 0.1;
}

// one step of the discrete family's environment (the kernels are instantiated once per family: ENV = AZG_ENV_CARTPOLE); Acrobot's
// dynamics live in include/azg_math.h, shared with the oracle
__device__ __forceinline__ void discrete_env_step(int env_id, const double* s, int action, double* o, double* reward, int* done) {
    if (env_id == AZG_ENV_MOUNTAINCAR) mountaincar_step(s, action, o, reward, done);
    else cartpole_step(s, action, o, reward, done);
}
// the same for a kernel of env family ENV: the Acrobot family steps Acrobot (include/azg_math.h), the CartPole family never does
template <int ENV>
__device__ __forceinline__ void family_env_step(int env_id, const double* s, int action, double* o, double* reward, int* done) {
    if constexpr (ENV == AZG_ENV_ACROBOT) azg_acrobot_step(s, action, o, reward, done);
    else discrete_env_step(env_id, s, action, o, reward, done);
}

// the discrete family's observation of a state as the network sees it: up to eight inputs (CartPole 4, MountainCar 2, Acrobot 6)
__device__ __forceinline__ void discrete_env_obs(int env_id, const double* s, float* obs8) {
    if (env_id == AZG_ENV_ACROBOT) { azg_acrobot_obs(s, obs8); obs8[6] = 0.0f; obs8[7] = 0.0f; }
    else { obs8[0] = (float)s[0]; obs8[1] = (float)s[1]; obs8[2] = (float)s[2]; obs8[3] = (float)s[3]; obs8[4] = obs8[5] = obs8[6] = obs8[7] = 0.0f; }
}

// Every step of the discrete family's environments that does not end the episode pays the same reward (CartPole +1, MountainCar -1,
// Acrobot -1): the tree walk takes a path record's reward from here instead of loading it (the node records still hold it, for
// dumps and the generic backup).  The step INTO a terminal node -- always a trace's last -- pays discrete_env_terminal_reward: the
// same for CartPole and MountainCar, 0 for Acrobot (gym: `reward = -1. if not terminal else 0.`).
__device__ __forceinline__ double discrete_env_reward(int env_id) { return env_id == AZG_ENV_CARTPOLE ? 1.0 : -1.0; }
__device__ __forceinline__ double discrete_env_terminal_reward(int env_id) {
    return env_id == AZG_ENV_CARTPOLE ? 1.0 : (env_id == AZG_ENV_ACROBOT ? 0.0 : -1.0);
}

// gym PendulumEnv.step; v1: speed clipped before integrating theta, v0: after.  sn_th = sin(theta), cached in the node.
// Two halves that share nothing but their inputs -- the new state, and the reward (a function of the OLD state and the action only) --
// so that a caller can take the reward later (tree_phases.cuh: DEFER); pendulum_step is both, in the order gym computes them.
__device__ __forceinline__ double pendulum_torque(float action) {
    const float max_torque = 2.0f;
    const float uc = action < -max_torque ? -max_torque : (action > max_torque ? max_torque : action);
    return (double)uc;
}
__device__ __forceinline__ double pendulum_reward(const double* s, float action) {
    const double pi = 3.141592653589793;
    const double th = s[0], thdot = s[1], u = pendulum_torque(action);
    const double an = azg_pymod(th + pi, 2.0 * pi, 0.15915494309189535) - pi;
    const double costs = (an * an + 0.1 * (thdot * thdot)) + 0.001 * (u * u);
    return -costs;
}
__device__ __forceinline__ void pendulum_dynamics(int v1, const double* s, double sn_th, float action, double* o) {
    const double max_speed = 8.0, dt = 0.05, pi = 3.141592653589793;
    const double th = s[0], thdot = s[1], u = pendulum_torque(action);
    double newth, newthdot, sn, cs;
    if (v1) {
        newthdot = thdot + (15.0 * sn_th + 3.0 * u) * dt;
        newthdot = newthdot < -max_speed ? -max_speed : (newthdot > max_speed ? max_speed : newthdot);
        newth = th + newthdot * dt;
    } else {
        azg_sincos(th + pi, &sn, &cs);
        newthdot = thdot + (-15.0 * sn + 3.0 * u) * dt;
        newth = th + newthdot * dt;
        newthdot = newthdot < -max_speed ? -max_speed : (newthdot > max_speed ? max_speed : newthdot);
    }
    o[0] = newth; o[1] = newthdot;
}
__device__ __forceinline__ void pendulum_step(int v1, const double* s, double sn_th, float action, double* o, double* reward, int* done) {
    *reward = pendulum_reward(s, action);
    pendulum_dynamics(v1, s, sn_th, action, o);
    *done = 0;
}
