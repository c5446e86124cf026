// np_motif.h -- the recognition sites of the four methylation alphabets (device): shared by the work-item builder (np_jobs_kernels.hip) and the
// genome-keyed site table (np_glue_kernels.hip)
#pragma once
#include <hip/hip_runtime.h>

namespace {
// The general form (recognition sites of up to 5 bases, up to two of them): nanopolish_alphabet.cpp:127-190, dam
// {"GATC","GMTC","CTMG"}, dcm {"CCAGG","CMAGG","GGTMC"} + {"CCTGG","CMTGG","GGAMC"}.  None of these sites can overlap another
// occurrence (no proper suffix of a site is a prefix of a site), so Alphabet::methylate's left-to-right scan marks exactly the
// FULL occurrences inside the string, and reverse_complement of the methylated string replaces exactly those by the
// methylated complements (a site cut by the string's end stays as it is: match_to_site needs the full length).
struct sites_t { int n, len; char s[2][5], m[2][5], mc[2][5]; };
__device__ __forceinline__ sites_t sites_of(int alphabet)
{
    sites_t t{};
    auto put = [&](int i, const char* a, const char* b, const char* c) { for (int q = 0; q < t.len; ++q) { t.s[i][q] = a[q]; t.m[i][q] = b[q]; t.mc[i][q] = c[q]; } };
    switch (alphabet) {
    case 2: t.n = 1; t.len = 2; put(0, "GC", "GM", "MG"); break;
    case 3: t.n = 1; t.len = 4; put(0, "GATC", "GMTC", "CTMG"); break;
    case 4: t.n = 2; t.len = 5; put(0, "CCAGG", "CMAGG", "GGTMC"); put(1, "CCTGG", "CMTGG", "GGAMC"); break;
    default: t.n = 1; t.len = 2; put(0, "CG", "MG", "GM"); break;
    }
    return t;
}
// index of the site that starts at position p of the window [w0, w0 + len) of ref and lies fully inside it, else -1
__device__ __forceinline__ int site_at(const char* __restrict__ ref, int w0, int len, int p, const sites_t& S)
{
    if (p < 0 || p + S.len > len) return -1;
    for (int i = 0; i < S.n; ++i) {
        bool ok = true;
        for (int q = 0; q < S.len; ++q) ok = ok && ref[w0 + p + q] == S.s[i][q];
        if (ok) return i;
    }
    return -1;
}
} // namespace
