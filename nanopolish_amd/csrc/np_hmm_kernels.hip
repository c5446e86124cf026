// np_hmm_kernels.hip -- profile-HMM forward (profile_hmm_score) and Viterbi (profile_hmm_align) for gfx950.
//
// Replaces profile_hmm_fill_generic_r9 (src/hmm/nanopolish_profile_hmm_r9.inl:265-433) and its two output
// writers (r9.inl:79-197).  Design (DESIGN.md section "Kernel B"):
//   * The lattice never exists in memory for the forward pass.  A job (e events x n k-mer blocks x 3 states)
//     is swept along anti-diagonals: lane j owns C consecutive k-mer blocks and, at step t, computes row
//     r = t - j of them.  Row r-1 of its own blocks is in its registers, rows r and r-1 of the block to its
//     left arrive through one DPP wave_shr:1 per state per step.  K(r,b) <- K(r,b-1) chains through the lane's
//     own C blocks inside the step and across lanes through the anti-diagonal skew.
//   * Jobs are short (typ. 38 x 16), so 64/SEG jobs share a wave in SEG-lane segments (SEG = 16/32/64).
//   * p7_FLogsum's 16000-entry table (64 KB) lives in LDS, one copy per 512-thread workgroup.
//   * Reduction order is the reference's: six terms in HMMMovementType order for M (r9.h:61-70), lp_end
//     accumulated row-ascending M,B,K by the lane that owns the last k-mer (r9.inl:388-396).
// All arithmetic is fp32 exactly as the reference; compile with -ffp-contract=off.
#include "np_kernels.h"
#include <type_traits>

#define NP_HMM_BLOCK 512

// timing experiments only (results WRONG with any bit set): 1 = emissions for free (the upper bound of what scoring a group's
// methylated and unmethylated sequence in one pass could share), 2 = log-sums without the table look-up (max only)
#ifndef NP_HMM_ABL
#define NP_HMM_ABL 0
#endif

namespace {

template <int C>
struct lane_state {
    float M[C], B[C], K[C];
};

__device__ __forceinline__ int wave_max_i32(int v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        int w = __shfl_xor(v, o, 64);
        v = v > w ? v : w;
    }
    return v;
}

// ---------------------------------------------------------------------------------------------------
// Forward
// ---------------------------------------------------------------------------------------------------
// BLK threads per workgroup share one LDS copy of the table.  One k-mer block per lane (C == 1) fits 64 VGPRs, so
// those classes run 1024-thread workgroups, two per CU = 8 waves/SIMD; the others run 512-thread workgroups.
#ifndef NP_HMM_FWD_BLOCK
#define NP_HMM_FWD_BLOCK 512
#endif
// OOR: the log-sum lookups rely on the LDS out-of-range rule (np_lse_oor) instead of clamping the index; np_create selects the
// clamped instantiation when its hardware probe fails (np_capi.hip:probe_hardware).
template <int SEG, int C, int BLK, bool OOR>
__global__ void __launch_bounds__(BLK, BLK / 128) np_hmm_forward_kernel(np_hmm_args a)
{
    __shared__ float tbl[NP_LOGSUM_TBL];          // the kernel's ONLY LDS object (np_lse_oor; checked at np_create)
    if (a.prio >= 2) __builtin_amdgcn_s_setprio(2);
    else if (a.prio == 1) __builtin_amdgcn_s_setprio(1);
    for (int i = threadIdx.x; i < NP_LOGSUM_TBL; i += BLK) tbl[i] = np_lse_table_entry(a.logsum, i);
    __syncthreads();
    const __attribute__((address_space(3))) char* tbl3 = (const __attribute__((address_space(3))) char*)tbl;
    auto NP_LSE = [&](float x, float y) -> float {
#if NP_HMM_ABL & 2
        return __builtin_fmaxf(x, y);
#endif
        if constexpr (OOR) return np_lse_oor(x, y, tbl3);
        else return np_lse(x, y, tbl);
    };

    constexpr int JPW = 64 / SEG;                 // jobs per wave
    const int lane = threadIdx.x & 63;
    const int seg = lane / SEG, sl = lane % SEG;
    const uint32_t n_jobs = *a.n_class_jobs;
    const uint32_t n_packs = (n_jobs + JPW - 1) / JPW;

    for (;;) {
        // (no `if (lane == 0)` around the atomic: see np_align_kernel.hip)
        const uint32_t pack = __builtin_amdgcn_readfirstlane(atomicAdd(a.counter, lane == 0 ? 1u : 0u));
        if (pack >= n_packs) break;

        const uint32_t slot = pack * JPW + seg;
        const bool has = seg < JPW && slot < n_jobs;          // (SEG = 3: lane 63 belongs to no segment)
        const uint32_t jidx = has ? a.order[slot] : 0u;
        const np_hmm_job_dev job = a.jobs[jidx];
        const np_read_dev* rd = a.reads + job.read;
        const int n = has ? (int)job.n_kmers : 0;
        const int e = has ? (int)(job.e_stop > job.e_start ? job.e_stop - job.e_start : job.e_start - job.e_stop) + 1 : 0;
        const int stride = job.stride;
        // k-mer blocks per lane: the wave runs every lane for cw = the largest ceil(n / SEG) among its items (the items
        // of a pack were binned by that number), instead of always C -- a 2-site CpG window of 18..26 k-mers on 4 lanes
        // then costs 5..7 block updates per step, not 8
        const int cw = __builtin_amdgcn_readfirstlane(wave_max_i32(has ? (n + SEG - 1) / SEG : 1));
        const int lanes_used = (n + cw - 1) / cw;
        const bool lane_on = has && sl < lanes_used;
        const float* ev = a.event_mean + rd->event_off;
        const bool pre_clip = (job.flags & NP_HAF_ALLOW_PRE_CLIP) != 0;
        const bool post_clip = (job.flags & NP_HAF_ALLOW_POST_CLIP) != 0;

        // per-read transitions (BlockTransitions, r9.h:75-95; identical for every k-mer, r9.inl:25-73)
        const float lp_mm_self = rd->trans[0], lp_mb = rd->trans[1], lp_mk = rd->trans[2], lp_mm_next = rd->trans[3],
                    lp_bb = rd->trans[4], lp_bk = rd->trans[5], lp_bm_next = rd->trans[6], lp_bm_self = rd->trans[7],
                    lp_kk = rd->trans[8], lp_km = rd->trans[9];

        // per-lane scaled Gaussians of this lane's k-mer blocks
        np_gauss g[C];
        {
            const double scale = rd->scale, shift = rd->shift, var = rd->var, log_var = rd->log_var;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int b = sl * cw + c;
                const uint32_t rank = (lane_on && c < cw && b < n) ? a.ranks[job.rank_off + b] : 0u;
                g[c] = np_scale_state(a.model, rank, scale, shift, var, log_var);
            }
        }

        lane_state<C> cur;
#pragma unroll
        for (int c = 0; c < C; ++c) cur.M[c] = cur.B[c] = cur.K[c] = NP_NEG_INF;   // row 0 (r9.cpp:21-33)
        float oM = NP_NEG_INF, oB = NP_NEG_INF, oK = NP_NEG_INF;   // left neighbour, row r-1
        float lp_end = NP_NEG_INF;
        float tM = NP_NEG_INF, tB = NP_NEG_INF, tK = NP_NEG_INF;   // the lane's last block, as of the row it computed last (row 0: -inf)
        const float head = sl == 0 ? NP_NEG_INF : 0.0f;
        auto shr_add = [&](const float v) {
            float r;
            asm("v_add_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(r) : "v"(v), "v"(head));
            return r;
        };
        const int last_lane = n > 0 ? (n - 1) / cw : 0, last_c = n > 0 ? (n - 1) % cw : 0;

        const int steps = __builtin_amdgcn_readfirstlane(wave_max_i32(has ? e + lanes_used - 1 : 0));      // (scalar: a uniform loop, not an exec-masked one)
        const bool is_last = lane_on && sl == last_lane;
        // software prefetch, one step ahead: event mean of the row this lane computes next, and the two clip-flank
        // values only the first / last k-mer's lane needs (pre_flank[r-1], post_flank[r-1] == flank[e-r])
        float xn = 0.0f, softn = NP_NEG_INF, pfn = 0.0f;
        if (lane_on && sl == 0) { xn = ev[job.e_start]; softn = a.flank[0]; }      // row 1: event_idx == e_start (r9.inl:361)
        if (is_last && sl == 0 && (post_clip || e == 1)) pfn = a.flank[e - 1];
        auto step = [&](const int t) __attribute__((always_inline)) {
            // left neighbour's row r (what lane j-1 computed in step t-1 for its last block, cw-1: a wave-uniform index);
            // segment heads see block 0 = -inf
            // (tM, tB, tK: the lane's LAST block, row r - 1 of the neighbour's next row.  Its index cw - 1 is wave-uniform but not a
            //  constant: picking it out of the eight candidates cost 21 selects per step; the block loop below leaves it in lM_r /
            //  lB_r / lK_r anyway, so the three values are carried from step to step instead -- round 4)
            // one instruction per state: the lane shift ADDS a per-lane constant -- -inf in a segment's first lane (block -1 = -inf), else
            // 0 (v + 0 == v for every value the lattice holds, v + -inf == -inf); bound_ctrl: lane 0's missing source reads as 0
            const float nM = shr_add(tM), nB = shr_add(tB), nK = shr_add(tK);

            const int r = t - sl;
            const bool act = lane_on && r >= 1 && r <= e;
            const float x = xn, soft = softn, pf = pfn;
            {
                const int rn = r + 1;                     // the row of the next step
                const bool actn = lane_on && rn >= 1 && rn <= e;
                xn = actn ? ev[job.e_start + (uint32_t)((rn - 1) * stride)] : 0.0f;                    // r9.inl:342
                softn = (actn && sl == 0 && (rn == 1 || pre_clip)) ? a.flank[rn - 1] : NP_NEG_INF;     // r9.inl:361-363
                pfn = (actn && is_last && (post_clip || rn == e)) ? a.flank[e - rn] : 0.0f;           // r9.inl:388
            }
            if (act) {
                float lM_r = nM, lB_r = nB, lK_r = nK;     // block to the left, row r
                float lM_p = oM, lB_p = oB, lK_p = oK;     // block to the left, row r-1
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    if (c >= cw) continue;                     // wave-uniform
#if NP_HMM_ABL & 1
                    const float em = x * g[c].cl;              // timing experiment only (scores WRONG): emissions for free
#else
                    const float em = np_emission(x, g[c]);
#endif
                    // PSR9_MATCH: HMT_FROM_SAME_M, PREV_M, SAME_B, PREV_B, PREV_K, SOFT (r9.inl:350-365)
                    float s = lp_mm_self + cur.M[c];
                    s = NP_LSE(s, lp_mm_next + lM_p);
                    s = NP_LSE(s, lp_bm_self + cur.B[c]);
                    s = NP_LSE(s, lp_bm_next + lB_p);
                    s = NP_LSE(s, lp_km + lK_p);
                    if (c == 0) s = NP_LSE(s, soft);   // HMT_FROM_SOFT: -inf except for the first k-mer
                    const float newM = s + em;
                    // PSR9_BAD_EVENT (r9.inl:368-374): only SAME_M and SAME_B are finite; emission 0
                    const float newB = NP_LSE(lp_mb + cur.M[c], lp_bb + cur.B[c]);
                    // PSR9_KMER_SKIP (r9.inl:377-383): PREV_M, PREV_B, PREV_K of the SAME row
                    const float newK = NP_LSE(NP_LSE(lp_mk + lM_r, lp_bk + lB_r), lp_kk + lK_r);

                    lM_p = cur.M[c]; lB_p = cur.B[c]; lK_p = cur.K[c];
                    lM_r = newM; lB_r = newB; lK_r = newK;
                    cur.M[c] = newM; cur.B[c] = newB; cur.K[c] = newK;

                    // end state (r9.inl:388-396): last k-mer, M then B then K
                    if (sl == last_lane && c == last_c && (post_clip || r == e)) {
                        lp_end = NP_LSE(lp_end, newM + pf);
                        lp_end = NP_LSE(lp_end, newB + pf);
                        lp_end = NP_LSE(lp_end, newK + pf);
                    }
                }
                tM = lM_r; tB = lB_r; tK = lK_r;               // what the last block of the loop left: row r of block cw - 1
            }
            oM = nM; oB = nB; oK = nK;
        };
        // two steps per iteration: what a step hands to the next (the rows, the neighbour's rows, the prefetched values) needs no move back
        // to fixed registers at the loop's back edge (27 moves per step in the one-step loop)
        int t = 1;
        for (; t + 1 <= steps; t += 2) { step(t); step(t + 1); }
        if (t <= steps) step(t);
        if (has && sl == last_lane) a.out[jidx] = lp_end;
    }
}

// ---------------------------------------------------------------------------------------------------
// Viterbi fill (ProfileHMMViterbiOutputR9, r9.inl:130-197): same sweep, max/arg-max with later-wins ties,
// lattice + back-pointers streamed to HBM in row-major [row][3*n] (block 0 / terminal block omitted).
// ---------------------------------------------------------------------------------------------------
struct vmax { float v; uint8_t from; };
__device__ __forceinline__ void vit_step(vmax& m, float x, uint8_t i)
{
    m.v = x > m.v ? x : m.v;
    m.from = (m.v == x) ? i : m.from;
}

template <int SEG, int C>
__global__ void __launch_bounds__(NP_HMM_BLOCK) np_hmm_viterbi_kernel(np_hmm_args a)
{
    constexpr int JPW = 64 / SEG;
    const int lane = threadIdx.x & 63;
    const int seg = lane / SEG, sl = lane % SEG;
    const uint32_t n_jobs = *a.n_class_jobs;
    const uint32_t n_packs = (n_jobs + JPW - 1) / JPW;

    for (;;) {
        // (no `if (lane == 0)` around the atomic: see np_align_kernel.hip)
        const uint32_t pack = __builtin_amdgcn_readfirstlane(atomicAdd(a.counter, lane == 0 ? 1u : 0u));
        if (pack >= n_packs) break;

        const uint32_t slot = pack * JPW + seg;
        const bool has = seg < JPW && slot < n_jobs;          // (SEG = 3: lane 63 belongs to no segment)
        const uint32_t jidx = has ? a.order[slot] : 0u;
        const np_hmm_job_dev job = a.jobs[jidx];
        const np_read_dev* rd = a.reads + job.read;
        const int n = has ? (int)job.n_kmers : 0;
        const int e = has ? (int)(job.e_stop > job.e_start ? job.e_stop - job.e_start : job.e_start - job.e_stop) + 1 : 0;
        const int stride = job.stride;
        const int lanes_used = (n + C - 1) / C;
        const bool lane_on = has && sl < lanes_used;
        const float* ev = a.event_mean + rd->event_off;
        const bool pre_clip = (job.flags & NP_HAF_ALLOW_PRE_CLIP) != 0;
        const int64_t cell0 = has ? a.cell_off[jidx] : 0;
        float* vm = a.vm + cell0;          // [(row-1)*3n + 3*b + state], rows 1..e
        uint8_t* bp = a.bp + cell0;
        const int rowlen = 3 * n;

        const float lp_mm_self = rd->trans[0], lp_mb = rd->trans[1], lp_mk = rd->trans[2], lp_mm_next = rd->trans[3],
                    lp_bb = rd->trans[4], lp_bk = rd->trans[5], lp_bm_next = rd->trans[6], lp_bm_self = rd->trans[7],
                    lp_kk = rd->trans[8], lp_km = rd->trans[9];

        np_gauss g[C];
        {
            const double scale = rd->scale, shift = rd->shift, var = rd->var, log_var = rd->log_var;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int b = sl * C + c;
                const uint32_t rank = (lane_on && b < n) ? a.ranks[job.rank_off + b] : 0u;
                g[c] = np_scale_state(a.model, rank, scale, shift, var, log_var);
            }
        }

        lane_state<C> cur;
#pragma unroll
        for (int c = 0; c < C; ++c) cur.M[c] = cur.B[c] = cur.K[c] = NP_NEG_INF;
        float oM = NP_NEG_INF, oB = NP_NEG_INF, oK = NP_NEG_INF;

        const int steps = wave_max_i32(has ? e + lanes_used - 1 : 0);
        for (int t = 1; t <= steps; ++t) {
            float nM = np_wave_shr1(cur.M[C - 1], NP_NEG_INF);
            float nB = np_wave_shr1(cur.B[C - 1], NP_NEG_INF);
            float nK = np_wave_shr1(cur.K[C - 1], NP_NEG_INF);
            if (sl == 0) { nM = NP_NEG_INF; nB = NP_NEG_INF; nK = NP_NEG_INF; }

            const int r = t - sl;
            const bool act = lane_on && r >= 1 && r <= e;
            if (act) {
                const uint32_t event_idx = job.e_start + (uint32_t)((r - 1) * stride);
                const float x = ev[event_idx];
                float lM_r = nM, lB_r = nB, lK_r = nK;
                float lM_p = oM, lB_p = oB, lK_p = oK;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const int b = sl * C + c;
                    const float em = np_emission(x, g[c]);
                    // MATCH: all six candidates in enum order, later index wins ties (r9.inl:138-143)
                    vmax m; m.v = lp_mm_self + cur.M[c]; m.from = 0;
                    vit_step(m, lp_mm_next + lM_p, 1);
                    vit_step(m, lp_bm_self + cur.B[c], 2);
                    vit_step(m, lp_bm_next + lB_p, 3);
                    vit_step(m, lp_km + lK_p, 4);
                    const float soft = (c == 0 && sl == 0 && (r == 1 || pre_clip)) ? a.flank[r - 1] : NP_NEG_INF;
                    vit_step(m, soft, 5);
                    const float newM = m.v + em;
                    // BAD_EVENT: x = [mb+M, -inf, bb+B, -inf, -inf, -inf]
                    vmax mb; mb.v = lp_mb + cur.M[c]; mb.from = 0;
                    vit_step(mb, NP_NEG_INF, 1);
                    vit_step(mb, lp_bb + cur.B[c], 2);
                    vit_step(mb, NP_NEG_INF, 3); vit_step(mb, NP_NEG_INF, 4); vit_step(mb, NP_NEG_INF, 5);
                    const float newB = mb.v + 0.0f;
                    // KMER_SKIP: x = [-inf, mk+M(r,b-1), -inf, bk+B(r,b-1), kk+K(r,b-1), -inf]
                    vmax mk; mk.v = NP_NEG_INF; mk.from = 0;
                    vit_step(mk, lp_mk + lM_r, 1);
                    vit_step(mk, NP_NEG_INF, 2);
                    vit_step(mk, lp_bk + lB_r, 3);
                    vit_step(mk, lp_kk + lK_r, 4);
                    vit_step(mk, NP_NEG_INF, 5);
                    const float newK = mk.v + 0.0f;

                    lM_p = cur.M[c]; lB_p = cur.B[c]; lK_p = cur.K[c];
                    lM_r = newM; lB_r = newB; lK_r = newK;
                    cur.M[c] = newM; cur.B[c] = newB; cur.K[c] = newK;

                    if (b < n) {
                        const int64_t o = (int64_t)(r - 1) * rowlen + 3 * b;
                        vm[o + 0] = newK; vm[o + 1] = newB; vm[o + 2] = newM;      // PSR9_KMER_SKIP=0, BAD_EVENT=1, MATCH=2
                        bp[o + 0] = mk.from; bp[o + 1] = mb.from; bp[o + 2] = m.from;
                    }
                }
            }
            oM = nM; oB = nB; oK = nK;
        }
    }
}

// Backtrack (profile_hmm_align_r9, r9.cpp:117-196): one lane per job walks the back-pointers from
// (last row, MATCH of last k-mer).  Output is written descending then reversed in place.
__global__ void np_hmm_backtrack_kernel(np_hmm_args a, int64_t n_jobs_total)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_jobs_total) return;
    const np_hmm_job_dev job = a.jobs[j];
    const int n = (int)job.n_kmers;
    if (n <= 0) { a.n_states[j] = 0; return; }
    const int e = (int)(job.e_stop > job.e_start ? job.e_stop - job.e_start : job.e_start - job.e_stop) + 1;
    if (e < 2) { a.n_states[j] = 0; return; }                 // assert(n_events >= 2), r9.cpp:88
    const float* vm = a.vm + a.cell_off[j];
    const uint8_t* bp = a.bp + a.cell_off[j];
    np_hmm_state* out = a.states + a.state_off[j];
    const int rowlen = 3 * n;
    int cnt = 0;
    bool bad = false;
    int row = e;                 // n_rows - 1
    int kmer = n - 1, ps = 2;    // col = 3*n_kmers + PSR9_MATCH  -> block n (k-mer n-1), MATCH
    while (row > 0) {
        if (kmer < 0) { bad = true; break; }                  // assert(block > 0)
        const int64_t o = (int64_t)(row - 1) * rowlen + 3 * kmer + ps;
        const float v = vm[o];
        if (v == NP_NEG_INF) { bad = true; break; }            // assert(get(vm,row,col) != -INFINITY)
        np_hmm_state st;
        st.event_idx = job.e_start + (uint32_t)((row - 1) * job.stride);
        st.kmer_idx = (uint32_t)kmer;
        st.l_fm = (double)v;
        st.state = "KBMNS"[ps];
        for (int q = 0; q < 7; ++q) st.pad[q] = 0;
        out[cnt++] = st;
        const int mv = bp[o];
        if (mv == 5) break;                                   // HMT_FROM_SOFT
        int next_ps = 2;
        switch (mv) {
            case 0: next_ps = 2; break;
            case 1: kmer -= 1; next_ps = 2; break;
            case 2: next_ps = 1; break;
            case 3: kmer -= 1; next_ps = 1; break;
            case 4: kmer -= 1; next_ps = 0; break;
        }
        if (ps != 0) row -= 1;                                // K states are silent (r9.cpp:176-178)
        ps = next_ps;
    }
    if (bad) { a.n_states[j] = 0; return; }
    for (int i = 0, k = cnt - 1; i < k; ++i, --k) { np_hmm_state t = out[i]; out[i] = out[k]; out[k] = t; }
    a.n_states[j] = cnt;
}


// ---------------------------------------------------------------------------------------------------
// Hardware probes run once per context (np_capi.hip:probe_hardware).  The forward kernel's clamp-free log-sum and the event
// aligner's / chain kernel's clamp-free prefetches rest on two hardware rules; a device, driver or compiler that breaks one
// must not produce silently wrong scores:
//   (1) an LDS read at or beyond the workgroup's allocation returns 0                      (np_lse_oor)
//   (2) a raw-buffer load outside [0, num_records) returns 0, a store there is dropped      (np_device.h: buf_f32, buf_store_u16)
// np_probe_dirty_kernel first fills every CU's whole LDS with non-zero words, so that (1) is not satisfied by leftovers.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) np_probe_dirty_kernel(int n_dwords, uint32_t* sink)
{
    extern __shared__ uint32_t dyn_lds[];
    for (int i = threadIdx.x; i < n_dwords; i += 1024) dyn_lds[i] = 0xdeadbeefu ^ (uint32_t)i;
    __syncthreads();
    if (dyn_lds[(threadIdx.x * 97) % n_dwords] == 0u) atomicAdd(sink, 1u);        // (keeps the stores alive)
}

// out[0]: np_lse_oor != np_lse on the probe pairs; out[1]: non-zero words read past the allocation; out[2]: buffer loads that
// broke rule (2); out[3]: workgroups that ran
__global__ void __launch_bounds__(NP_HMM_FWD_BLOCK) np_probe_kernel(const float* logsum, const float* buf, uint16_t* sbuf, uint32_t* out)
{
    __shared__ float tbl[NP_LOGSUM_TBL];          // the same single LDS object as np_hmm_forward_kernel
    for (int i = threadIdx.x; i < NP_LOGSUM_TBL; i += NP_HMM_FWD_BLOCK) tbl[i] = np_lse_table_entry(logsum, i);
    __syncthreads();
    const __attribute__((address_space(3))) char* tbl3 = (const __attribute__((address_space(3))) char*)tbl;
    const uint32_t base = (uint32_t)(uintptr_t)tbl3;
    uint32_t bad_lds = 0, bad_lse = 0, bad_buf = 0;
    // (1) every word from the end of the allocation to past the end of the CU's LDS (160 KB), and the far offsets the
    //     saturated conversion produces
    for (uint32_t o = NP_LOGSUM_TBL * 4u + 4u * threadIdx.x; o < 168u * 1024u; o += 4u * NP_HMM_FWD_BLOCK)
        bad_lds += *(const volatile __attribute__((address_space(3))) uint32_t*)(uintptr_t)(base + o) != 0u;
    {
        const uint32_t far[6] = {0x00100000u, 0x01000000u, 0x7ffffffcu, 0x80000000u, 0xfffffff0u, 0xfffffffcu};
        for (int i = 0; i < 6; ++i) bad_lds += *(const volatile __attribute__((address_space(3))) uint32_t*)(uintptr_t)(base + far[i]) != 0u;
    }
    // the two lookups must agree bit for bit wherever the forward pass can take them
    {
        const float inf = __builtin_inff();
        const float d[16] = {0.0f, 0.00099f, 0.001f, 7.3f, 15.69f, 15.6995f, 15.7f, 15.9999f, 16.0f, 16.0001f, 17.5f, 100.0f, 1.0e6f, 4.0e9f, 1.0e30f, inf};
        const float x = -1.0f - 0.37f * (float)threadIdx.x;
        for (int i = 0; i < 16; ++i) {
            const float y = x - d[i];
            bad_lse += __builtin_bit_cast(uint32_t, np_lse_oor(x, y, tbl3)) != __builtin_bit_cast(uint32_t, np_lse(x, y, tbl));
            bad_lse += __builtin_bit_cast(uint32_t, np_lse_oor(y, x, tbl3)) != __builtin_bit_cast(uint32_t, np_lse(y, x, tbl));
        }
        bad_lse += np_lse_oor(-inf, -inf, tbl3) != -inf;
        bad_lse += np_lse_oor(-inf, x, tbl3) != x;
    }
    // (2) a descriptor over the first 64 bytes of buf (sixteen 1.0f): in range reads 1, everything else 0
    if (threadIdx.x < 64) {
        const __amdgpu_buffer_rsrc_t r = make_rsrc(buf, 64u);
        const int oob[8] = {-4, -64, 64, 68, 1 << 20, 0x7ffffff0, (int)0x80000000u, (int)0xc0000000u};
        for (int i = 0; i < 8; ++i) bad_buf += buf_f32(r, oob[i]) != 0.0f;
        bad_buf += buf_f32(r, 4 * (threadIdx.x & 15)) != 1.0f;
        // stores: the descriptor covers the first 32 bytes of sbuf; the host checks that the 32 bytes after them kept their pattern
        const __amdgpu_buffer_rsrc_t w = make_rsrc(sbuf, 32u);
        buf_store_u16(w, 32 + 2 * (threadIdx.x & 15), 0xffffu);
        buf_store_u16(w, -2 - 2 * (int)(threadIdx.x & 15), 0xffffu);
        if (blockIdx.x == 0 && threadIdx.x < 16) buf_store_u16(w, 2 * threadIdx.x, 0x1234u);      // in range: must land
    }
    if (bad_lse) atomicAdd(out + 0, bad_lse);
    if (bad_lds) atomicAdd(out + 1, bad_lds);
    if (bad_buf) atomicAdd(out + 2, bad_buf);
    if (threadIdx.x == 0) atomicAdd(out + 3, 1u);
}

template <int SEG, int C>
hipError_t launch_fwd(const np_hmm_args& a, int n_blocks, bool oor, hipStream_t s)
{
    constexpr int BLK = NP_HMM_FWD_BLOCK;
    if (oor) hipLaunchKernelGGL((np_hmm_forward_kernel<SEG, C, BLK, true>), dim3(n_blocks), dim3(BLK), 0, s, a);
    else hipLaunchKernelGGL((np_hmm_forward_kernel<SEG, C, BLK, false>), dim3(n_blocks), dim3(BLK), 0, s, a);
    return hipGetLastError();
}
// static LDS bytes of the OOR instantiation of a size class: must be exactly the table (np_lse_oor's precondition)
template <int SEG, int C>
hipError_t fwd_lds_bytes(size_t* bytes)
{
    hipFuncAttributes at;
    const hipError_t e = hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&np_hmm_forward_kernel<SEG, C, NP_HMM_FWD_BLOCK, true>));
    if (e == hipSuccess) *bytes = at.sharedSizeBytes;
    return e;
}
template <int SEG, int C>
hipError_t launch_vit(const np_hmm_args& a, int n_blocks, hipStream_t s)
{
    hipLaunchKernelGGL((np_hmm_viterbi_kernel<SEG, C>), dim3(n_blocks), dim3(NP_HMM_BLOCK), 0, s, a);
    return hipGetLastError();
}

} // namespace

int np_hmm_block_threads(int cls) { (void)cls; return NP_HMM_FWD_BLOCK; }
int np_vit_block_threads(void) { return NP_HMM_BLOCK; }

hipError_t np_launch_hmm_forward(int cls, const np_hmm_args& a, int n_blocks, bool lse_oor, hipStream_t s)
{
    switch (cls) {
        case 0: return launch_fwd<2, 8>(a, n_blocks, lse_oor, s);
        case 1: return launch_fwd<3, 8>(a, n_blocks, lse_oor, s);
        case 2: return launch_fwd<4, 8>(a, n_blocks, lse_oor, s);
        case 3: return launch_fwd<8, 8>(a, n_blocks, lse_oor, s);
        case 4: return launch_fwd<16, 8>(a, n_blocks, lse_oor, s);
        case 5: return launch_fwd<32, 8>(a, n_blocks, lse_oor, s);
        case 6: return launch_fwd<64, 8>(a, n_blocks, lse_oor, s);
        case 7: return launch_fwd<64, 16>(a, n_blocks, lse_oor, s);
    }
    return hipErrorInvalidValue;
}

hipError_t np_launch_probe(const float* logsum, const float* buf, uint16_t* sbuf, uint32_t* out, int n_blocks, hipStream_t s)
{
    // best effort: dirty the whole LDS of every CU first (a launch that cannot get 160 KB of dynamic LDS is simply skipped)
    const int lds_bytes = 160 * 1024;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&np_probe_dirty_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) == hipSuccess) {
        hipLaunchKernelGGL(np_probe_dirty_kernel, dim3(n_blocks), dim3(1024), lds_bytes, s, lds_bytes / 4, out + 4);
        (void)hipGetLastError();
    }
    hipFuncAttributes at;
    hipError_t e = hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&np_probe_kernel));
    if (e != hipSuccess) return e;
    if (at.sharedSizeBytes != NP_LOGSUM_TBL * sizeof(float)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(np_probe_kernel, dim3(n_blocks), dim3(NP_HMM_FWD_BLOCK), 0, s, logsum, buf, sbuf, out);
    return hipGetLastError();
}

hipError_t np_hmm_forward_lds_bytes(int cls, size_t* bytes)
{
    switch (cls) {
        case 0: return fwd_lds_bytes<2, 8>(bytes);
        case 1: return fwd_lds_bytes<3, 8>(bytes);
        case 2: return fwd_lds_bytes<4, 8>(bytes);
        case 3: return fwd_lds_bytes<8, 8>(bytes);
        case 4: return fwd_lds_bytes<16, 8>(bytes);
        case 5: return fwd_lds_bytes<32, 8>(bytes);
        case 6: return fwd_lds_bytes<64, 8>(bytes);
        case 7: return fwd_lds_bytes<64, 16>(bytes);
    }
    return hipErrorInvalidValue;
}

hipError_t np_launch_hmm_viterbi(int cls, const np_hmm_args& a, int n_blocks, hipStream_t s)
{
    switch (cls) {
        case 0: return launch_vit<2, 8>(a, n_blocks, s);
        case 1: return launch_vit<3, 8>(a, n_blocks, s);
        case 2: return launch_vit<4, 8>(a, n_blocks, s);
        case 3: return launch_vit<8, 8>(a, n_blocks, s);
        case 4: return launch_vit<16, 8>(a, n_blocks, s);
        case 5: return launch_vit<32, 8>(a, n_blocks, s);
        case 6: return launch_vit<64, 8>(a, n_blocks, s);
        case 7: return launch_vit<64, 16>(a, n_blocks, s);
    }
    return hipErrorInvalidValue;
}

hipError_t np_launch_hmm_backtrack(const np_hmm_args& a, int64_t n_jobs, hipStream_t s)
{
    if (n_jobs <= 0) return hipSuccess;
    const int bs = 64;
    hipLaunchKernelGGL(np_hmm_backtrack_kernel, dim3((unsigned)((n_jobs + bs - 1) / bs)), dim3(bs), 0, s, a, n_jobs);
    return hipGetLastError();
}
