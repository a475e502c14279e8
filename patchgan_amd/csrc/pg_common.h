// pg_common.h -- device helpers shared by the HIP translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include "patchgan_hip.h"

__device__ __forceinline__ float pg_act(float v, int act) {
    switch (act) {
        case PG_ACT_LEAKY: return v > 0.f ? v : 0.2f * v;
        case PG_ACT_RELU: return v > 0.f ? v : 0.f;
        case PG_ACT_TANH: return tanhf(v);
        case PG_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        default: return v;
    }
}

// derivative of the activation expressed through its OUTPUT a (what torch's in-place activations save)
__device__ __forceinline__ float pg_act_grad_from_out(float a, int act) {
    switch (act) {
        case PG_ACT_LEAKY: return a > 0.f ? 1.f : 0.2f;
        case PG_ACT_RELU: return a > 0.f ? 1.f : 0.f;
        case PG_ACT_TANH: return 1.f - a * a;
        case PG_ACT_SIGMOID: return a * (1.f - a);
        default: return 1.f;
    }
}

// the same values without a switch (selects only): for epilogues unrolled over many accumulators, where branches cost registers
__device__ __forceinline__ float pg_act_grad_sel(float a, int act) {
    const float pw = a > 0.f ? 1.f : (act == PG_ACT_LEAKY ? 0.2f : 0.f);
    const float sm = act == PG_ACT_TANH ? 1.f - a * a : a * (1.f - a);
    return act == PG_ACT_NONE ? 1.f : ((act == PG_ACT_LEAKY || act == PG_ACT_RELU) ? pw : sm);
}

// Counter-based dropout RNG: keep element e of stream `seed` iff a 64-bit mix of (seed, e) maps to u >= p.
__device__ __forceinline__ uint64_t pg_mix64(uint64_t x) {
    x ^= x >> 30;
    x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27;
    x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return x;
}
__device__ __forceinline__ bool pg_dropout_keep(uint64_t seed, uint64_t e, float p) {
    const uint64_t h = pg_mix64(seed + 0x9e3779b97f4a7c15ULL * (e + 1));
    const float u = (float)(h >> 40) * (1.0f / 16777216.0f);   // 24 random bits -> [0,1)
    return u >= p;
}

// XCD-aware work order (cdna_hip_programming.md T1): workgroups b and b+8 share an XCD (round-robin dispatch of the
// flattened block id), so map block id -> work index such that each XCD walks a CONTIGUOUS run of work indices: neighbouring
// work items (same operand slab) then hit that XCD's private 4 MiB L2 instead of the fabric.  Bijective for any count; speed only.
__device__ __forceinline__ int pg_xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, k = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// optional last step of a data-gradient epilogue: out *= f'(t[pixel * ld + c]), f' expressed through the activation OUTPUT t
// (pg_act_grad_from_out) -- the standalone activation-backward pass folded into the kernel that produces its input.  t == nullptr: none.
// t has the storage type of the output tensor.
struct pg_epi_mul {
    const void* t;
    int ld, act;
};

// The PATCHGAN_* tuning variables (tile / split sweeps, kernel-family switches) exist for same-device A/B timing and
// are honoured only when PATCHGAN_EXPERIMENT is set: a production process reads exactly one environment variable on this path.
// (Callers and tests select kernels through the PG_TUNE_* bits of `algo`.)
static inline const char* pg_exp_env(const char* name) {
    static const bool on = getenv("PATCHGAN_EXPERIMENT") != nullptr;
    return on ? getenv(name) : nullptr;
}

static inline int pg_launch_status() { return hipGetLastError() == hipSuccess ? PG_OK : PG_ELAUNCH; }
