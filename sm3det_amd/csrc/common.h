// common.h -- helpers shared by the gfx950 translation units of libsm3det_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/sm3det_hip.h"

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Launch errors only (no synchronisation): the C ABI never blocks the host.
static inline int launch_status() { return hipGetLastError() == hipSuccess ? SM3_OK : SM3_ERR_LAUNCH; }

constexpr int kNumXCD = 8;  // MI355X: 8 XCDs x 32 CUs, block b runs on XCD b % 8 (speed only, never correctness)
constexpr int kNumCU = 256;
