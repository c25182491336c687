// Deterministic mode (tedspad_set_deterministic, include/tedspad_hip.h): the float-atomic sections of the training kernels -- BatchNorm batch statistics in
// the conv epilogues, channel sums of the BatchNorm backward, the weight-gradient flush -- are passed by the workgroups of a launch ONE AT A TIME, in
// blockIdx order, so that every accumulator receives its addends in the same order on every run. Off (the default) the gate is one uniform load per workgroup.
//
// Protocol: a ticket counts the gate passages of the running launch; workgroup b (its `sub`-th passage of `nsub`) waits for ticket == b * nsub + sub, adds,
// fences, and publishes ticket + 1; the launch's last passage puts it back to 0 for the next launch. Workgroups are dispatched in blockIdx order, so the
// workgroup the others wait for is always resident or next in line: no deadlock. A workgroup that still does not see its turn after ~2^22 polls sets a
// "gave up" flag that opens the gate for everybody (results are then merely not reproducible) -- a stuck wave must never outlive its launch on this pool;
// tedspad_deterministic_giveups() reports it. The state is per translation unit (an anonymous-namespace __device__ array), i.e. per kernel family; two
// launches of one family never run concurrently in deterministic mode (train_engine keeps the weight gradients on the main stream then).
#pragma once
#include "common.h"

namespace tedspad {
namespace {

__device__ int g_det[4];      // [0] on, [1] ticket, [2] workgroups that gave up

__device__ __forceinline__ unsigned det_block() { return blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); }

// Call from workgroup-uniform code (it contains a barrier). Returns whether the gate is on (pass that to det_exit).
__device__ __forceinline__ bool det_enter(int sub = 0, int nsub = 1) {
    const int on = __builtin_amdgcn_readfirstlane(g_det[0]);
    if (!on) return false;
    if (threadIdx.x == 0) {
        const int my = (int)(det_block() * (unsigned)nsub + (unsigned)sub);
        int polls = 0;
        while (__hip_atomic_load(&g_det[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != my &&
               __hip_atomic_load(&g_det[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __builtin_amdgcn_s_sleep(32);
            if (++polls > (1 << 22)) {
                __hip_atomic_fetch_add(&g_det[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
    return true;
}

__device__ __forceinline__ void det_exit(bool on, int sub = 0, int nsub = 1) {
    if (!on) return;
    __threadfence();                      // this workgroup's adds are performed before the next one is let in
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned my = det_block() * (unsigned)nsub + (unsigned)sub;
        const unsigned total = gridDim.x * gridDim.y * gridDim.z * (unsigned)nsub;
        __hip_atomic_store(&g_det[1], my + 1 == total ? 0 : (int)(my + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// host side of this translation unit's copy: op 0 = set (on / off, ticket and flag cleared), op 1 = read the give-up count
inline int32_t det_ctl(int op, int on) {
    int v[4] = {on, 0, 0, 0};
    if (op == 0) return hipMemcpyToSymbol(HIP_SYMBOL(g_det), v, sizeof(v)) == hipSuccess ? 0 : -1;
    if (hipMemcpyFromSymbol(v, HIP_SYMBOL(g_det), sizeof(v)) != hipSuccess) return -1;
    return v[2];
}

}  // namespace

// one per kernel family (defined in conv_igemm.hip, conv_patch.hip, conv_wgrad.hip, train_ops.hip)
int32_t det_ctl_igemm(int op, int on);
int32_t det_ctl_patch(int op, int on);
int32_t det_ctl_wgrad(int op, int on);
int32_t det_ctl_train_ops(int op, int on);

}  // namespace tedspad
