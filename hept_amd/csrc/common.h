// Shared device/host helpers for the gfx950 HEPT kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hept_hip.h"

#define HEPT_WAVE 64

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

static inline int hept_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? HEPT_OK : HEPT_ERR_LAUNCH;
}

// float -> bf16 bits, round to nearest even (inputs are finite here).
__device__ __forceinline__ unsigned int hept_bf16_bits(float x) {
    unsigned int u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float hept_bf16_round(float x) {
    return __uint_as_float(hept_bf16_bits(x) << 16);
}
__device__ __forceinline__ unsigned int hept_pack_bf16(float lo, float hi) {
    return hept_bf16_bits(lo) | (hept_bf16_bits(hi) << 16);
}

// Row of the 32x32 MFMA accumulator held in register r by lane-half hh
// (C/D layout of v_mfma_f32_32x32x*: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)).
__device__ __forceinline__ int hept_acc_row(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }
