// ke_lz_copies.h -- copies from earlier output, made by one wave, 64 at a time: the LZ77 matches of a deflate stream (ke_png.hip)
// and the dictionary strings of an LZW stream (ke_gif.hip: a string is the previous one plus a character, and both lie in the
// output already).  The stream's reader has written the literals to their places and recorded every copy as
// {destination, distance << 9 | (length - bias)}; this makes the copies in order.
#pragma once

#include <stdint.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 ld16(const uint8_t *p) {       // any alignment
    u32x4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}
__device__ __forceinline__ void st16(uint8_t *p, u32x4 v) { __builtin_memcpy(p, &v, 16); }

// ---- the LZ77 copies, one wave per image, 64 at a time.  A copy can go as soon as the bytes it reads are final: everything
// below the destination of the first copy not yet made is (literals are in place, earlier copies are done).  Copies of one
// round read below that mark and write at or above it, so they do not touch each other; a copy that overlaps its own output
// (distance < length) repeats its first `distance` bytes.  The first pending copy always qualifies, so every round makes
// progress, and in the common case -- sources further back than the 64 copies span -- one round does all 64.

__device__ __forceinline__ void store_low(uint8_t *d, uint64_t lo, uint64_t hi, uint32_t rem) {      // rem < 16 bytes of lo:hi
    if (rem & 8) { __builtin_memcpy(d, &lo, 8); d += 8; lo = hi; }
    if (rem & 4) { const uint32_t v = (uint32_t)lo; __builtin_memcpy(d, &v, 4); d += 4; lo >>= 32; }
    if (rem & 2) { const uint16_t v = (uint16_t)lo; __builtin_memcpy(d, &v, 2); d += 2; lo >>= 16; }
    if (rem & 1) *d = (uint8_t)lo;
}

// One copy: `len` bytes at d.  src: where its pattern comes from -- d - dist for a copy on its own; for a copy that continues
// a run of the same period (see the kernel) the bytes in front of the run's first copy, with `phase` = this copy's offset into
// the period.
__device__ __forceinline__ void copy_match(uint8_t *d, const uint8_t *src, uint32_t dist, uint32_t len, uint32_t phase) {
    if (dist > 16) {
        const uint8_t *s = d - dist;
        if (len >= 16) {
            uint32_t k = 0;
            for (; k + 16 <= len; k += 16) st16(d + k, ld16(s + k));
            if (k < len) st16(d + len - 16, ld16(s + len - 16));      // the last 16 bytes again, ending where the copy ends
        } else {
            const u32x4 v = ld16(s);
            store_low(d, (uint64_t)v.x | ((uint64_t)v.y << 32), (uint64_t)v.z | ((uint64_t)v.w << 32), len);
        }
        return;
    }
    // the 16-byte pattern of period `dist`: keep the first dist bytes, double them up
    const u32x4 v = ld16(src);
    uint64_t lo = (uint64_t)v.x | ((uint64_t)v.y << 32), hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
    if (dist < 8) { lo &= (1ull << (8 * dist)) - 1ull; hi = 0; } else if (dist < 16) { hi &= (1ull << (8 * (dist - 8))) - 1ull; }
    for (uint32_t q = dist; q < 16; q <<= 1) {
        uint64_t nl, nh;
        if (q < 8) { nh = (hi << (8 * q)) | (lo >> (64 - 8 * q)); nl = lo << (8 * q); } else { nh = lo << (8 * (q - 8)); nl = 0; }
        lo |= nl;
        hi |= nh;
    }
    if (phase) {                                       // start the pattern `phase` bytes into its period
        uint64_t rlo = 0, rhi = 0;
        uint32_t r = phase;
        for (int k = 0; k < 16; ++k) {
            const uint64_t byte = ((r < 8 ? lo >> (8 * r) : hi >> (8 * (r - 8))) & 255ull);
            if (k < 8) rlo |= byte << (8 * k); else rhi |= byte << (8 * (k - 8));
            r = r + 1 == dist ? 0 : r + 1;
        }
        lo = rlo;
        hi = rhi;
    }
    const u32x4 pat{(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
    // largest multiple of dist within 16, minus one, for dist 1..16: stepping by it keeps the pattern in phase
    const uint32_t step = dist == 16 ? 16u : 1u + (uint32_t)((0xEDCBA98FDBEFEFFull >> (4 * (dist - 1))) & 15u);
    uint32_t k = 0;
    for (; k + 16 <= len; k += step) st16(d + k, pat);
    store_low(d + k, lo, hi, len - k);
}

// number of lanes whose `sorted` value (ascending over the lanes) is <= x (inclusive) or < x
__device__ __forceinline__ int lanes_below(uint32_t sorted, uint32_t x, bool inclusive) {
    int lo = 0;
#pragma unroll
    for (int half = 32; half >= 1; half >>= 1) {
        const uint32_t v = (uint32_t)__shfl((int)sorted, lo + half - 1);
        lo += (inclusive ? v <= x : v < x) ? half : 0;
    }
    // lanes 0 .. lo-1 qualify so far; the search above covers 63 lanes, the last one is checked on its own
    const uint32_t v = (uint32_t)__shfl((int)sorted, lo < 64 ? lo : 63);
    if (lo == 63 && (inclusive ? v <= x : v < x)) lo = 64;
    return lo;
}

// the `m` recorded copies of one stream whose output starts at p; all 64 lanes call it
__device__ __forceinline__ void ke_lz_make_copies(uint8_t *p, const uint2 *rec, uint32_t m, uint32_t len_bias) {
    const int lane = threadIdx.x;
    for (uint32_t first = 0; first < m; first += 64) {
        const uint32_t j = first + lane;
        bool pending = j < m;
        uint32_t dst = 0xFFFFFFFFu, dist = 1, len = 0;              // lanes beyond the list sort behind everything
        if (pending) {
            const uint2 r = rec[j];
            dst = r.x; dist = r.y >> 9; len = (r.y & 511u) + len_bias;
        }
        const uint32_t end = pending ? dst + len : 0xFFFFFFFFu;
        // Runs: a copy that starts where the one before it ends, with the same distance of at most 16, continues that copy's
        // periodic pattern -- its bytes follow from the bytes in front of the RUN's first copy, not from its neighbour's
        // output, so a whole run (a flat area is thousands of such copies) can go in one round.
        const uint32_t prev_end = (uint32_t)__shfl_up((int)end, 1), prev_dist = (uint32_t)__shfl_up((int)dist, 1);
        const bool continues = lane > 0 && pending && dist <= 16 && dist == prev_dist && dst == prev_end;
        const uint64_t heads = __ballot(!continues);
        const int head = 63 - __clzll((long long)(heads & (~0ull >> (63 - lane))));
        const uint32_t head_dst = (uint32_t)__shfl((int)dst, head);
        const uint32_t phase = continues ? (dst - head_dst) % dist : 0;
        // what has to be final before this copy can go: [from, to), the bytes in front of its run (its own first `dist`
        // source bytes when it stands alone); the copies whose destinations reach into that range are lanes [lo, hi)
        const uint32_t from = head_dst - dist, to = from + (continues ? dist : min(dist, len));
        const int lo = lanes_below(end, from, true), hi = lanes_below(dst, to, false);
        const uint64_t blockers = hi > lo ? ((hi - lo >= 64 ? ~0ull : (1ull << (hi - lo)) - 1ull) << lo) : 0ull;
        for (;;) {
            const uint64_t waiting = __ballot(pending);
            if (!waiting) break;
            if (pending && (waiting & blockers) == 0) {
                copy_match(p + dst, p + from, dist, len, phase);
                pending = false;
            }
        }
    }
}
