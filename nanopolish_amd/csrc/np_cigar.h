// np_cigar.h -- device-side view of a BAM CIGAR without materialising its aligned pairs (shared by the work-item generator,
// np_jobs_kernels.hip, and the eventalign chain, np_eventalign_kernel.hip).
// get_aligned_segments (src/alignment/nanopolish_anchor.cpp:20-95) emits one (ref_pos, read_pos) pair per base of every M/=/X
// operation; both coordinates grow along the pairs, so any lower/upper bound over them is one binary search over the
// operations' scanned offsets (np_cigar_index_kernel) plus a short walk over neighbouring non-aligned operations.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

// ---- CIGAR view ---------------------------------------------------------------------------------------------------------
struct cig_view {
    const uint32_t* cigar;     // BAM words: length << 4 | op
    const int32_t* op_ref;     // reference offset (relative to the record's pos) at the start of every operation, n + 1 entries
    const int32_t* op_read;    // read offset (reference strand) likewise
    int n;
};
__device__ __forceinline__ bool op_aligned(uint32_t w) { const uint32_t op = w & 0xf; return op == 0 || op == 7 || op == 8; }

// first aligned pair with ref_pos >= x: the operation that contains x is the last one starting at or before it
__device__ __forceinline__ bool first_aligned_ref_ge(const cig_view& c, int x, int& q, int& r)
{
    int lo = 0, hi = c.n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (c.op_ref[mid] <= x) lo = mid + 1; else hi = mid; }
    for (int i = lo > 0 ? lo - 1 : 0; i < c.n; ++i) {
        const uint32_t w = c.cigar[i];
        const int len = (int)(w >> 4);
        if (!op_aligned(w) || len == 0) continue;
        const int off = x - c.op_ref[i] > 0 ? x - c.op_ref[i] : 0;
        if (off < len) { q = c.op_read[i] + off; r = c.op_ref[i] + off; return true; }
    }
    return false;
}
// first aligned pair with read_pos >= y
__device__ __forceinline__ bool first_aligned_read_ge(const cig_view& c, int y, int& q, int& r)
{
    int lo = 0, hi = c.n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (c.op_read[mid] <= y) lo = mid + 1; else hi = mid; }
    for (int i = lo > 0 ? lo - 1 : 0; i < c.n; ++i) {
        const uint32_t w = c.cigar[i];
        const int len = (int)(w >> 4);
        if (!op_aligned(w) || len == 0) continue;
        const int off = y - c.op_read[i] > 0 ? y - c.op_read[i] : 0;
        if (off < len) { q = c.op_read[i] + off; r = c.op_ref[i] + off; return true; }
    }
    return false;
}
// last aligned pair with read_pos <= y
__device__ __forceinline__ bool last_aligned_read_le(const cig_view& c, int y, int& q)
{
    int lo = 0, hi = c.n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (c.op_read[mid] <= y) lo = mid + 1; else hi = mid; }
    for (int i = lo - 1; i >= 0; --i) {
        const uint32_t w = c.cigar[i];
        const int len = (int)(w >> 4);
        if (!op_aligned(w) || len == 0) continue;
        const int off = y - c.op_read[i] < len - 1 ? y - c.op_read[i] : len - 1;
        if (off >= 0) { q = c.op_read[i] + off; return true; }
    }
    return false;
}


// last aligned pair with read_pos <= y, with its reference offset
__device__ __forceinline__ bool last_aligned_read_le_r(const cig_view& c, int y, int& q, int& r)
{
    int lo = 0, hi = c.n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (c.op_read[mid] <= y) lo = mid + 1; else hi = mid; }
    for (int i = lo - 1; i >= 0; --i) {
        const uint32_t w = c.cigar[i];
        const int len = (int)(w >> 4);
        if (!op_aligned(w) || len == 0) continue;
        const int off = y - c.op_read[i] < len - 1 ? y - c.op_read[i] : len - 1;
        if (off >= 0) { q = c.op_read[i] + off; r = c.op_ref[i] + off; return true; }
    }
    return false;
}
// last aligned pair with ref_pos <= x
__device__ __forceinline__ bool last_aligned_ref_le(const cig_view& c, int x, int& q, int& r)
{
    int lo = 0, hi = c.n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (c.op_ref[mid] <= x) lo = mid + 1; else hi = mid; }
    for (int i = lo - 1; i >= 0; --i) {
        const uint32_t w = c.cigar[i];
        const int len = (int)(w >> 4);
        if (!op_aligned(w) || len == 0) continue;
        const int off = x - c.op_ref[i] < len - 1 ? x - c.op_ref[i] : len - 1;
        if (off >= 0) { q = c.op_read[i] + off; r = c.op_ref[i] + off; return true; }
    }
    return false;
}
