// BLAKE2b-512 (RFC 7693) on the host CPU: setup_utils::calculate_hash (reference call sites
// /root/reference/src/bin/verify_transcript.rs:344-345, src/utils.rs:618-623).  The hash chain of a chunk is inherently
// sequential (a 503 MB challenge is ONE BLAKE2b stream) and bounds the wall time of a file-level call, so the compression
// function has an AVX2 form (the four columns / diagonals of a round in 256-bit registers), chosen at run time; the portable
// form is the fallback and the checker (tests compare both with hashlib).
#pragma once
#include <cstddef>
#include <cstdint>

namespace ptau {

struct Blake2b {
    uint64_t h[8], t0 = 0, t1 = 0;
    uint8_t buf[128];
    size_t fill = 0;
    Blake2b();
    void update(const uint8_t *p, size_t n);
    void finish(uint8_t out[64]);
};

// 0 = portable, 1 = AVX2; `force` < 0 leaves the run-time choice alone (test hook)
int blake2b_implementation(int force);

}  // namespace ptau
using ptau::Blake2b;
